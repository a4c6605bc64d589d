R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for W in C5M C5; do for P in 3 4 5 6 8; do echo -n "$W $P parts: "; run --workload $W --wave-parts $P; done; done
for B in 8192 16384; do for P in 3 4; do echo -n "C3 $B books $P parts: "; run --books $B --wave-parts $P --steps 200 --warmup 50; done; done
for B in 4096 16384; do for P in 2 3 4; do echo -n "C5M shape $B books $P parts: "; run --workload C5M --books $B --wave-parts $P; done; done
