R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for B in 4096 6144 8192 12288 16384 24576; do
  for P in auto wave_persist; do echo -n "C3 $B books $P: "; run --books $B --pipeline $P; done
done
for B in 4096 8192 16384; do for P in auto wave_persist; do echo -n "C2 $B books $P: "; run --workload C2 --books $B --pipeline $P; done; done
