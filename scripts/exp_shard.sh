# Round-4 experiments at the C4 shard sizes (GPU box): parts sweep, lazy cancellations.   bash scripts/exp_shard.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s  %s' % (d['value']/1e6, d['runs']['median']/1e6, d['config']['pipeline'][:40], {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for B in 8192 16384; do
  for P in 2 3 4 6; do echo -n "books $B wave_parts $P: "; run --books $B --pipeline wave_split --wave-parts $P; done
  echo -n "books $B lazy-cancel, 3 parts:   "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_lazy.so run --books $B
done
echo -n "C3 base: "; run
echo -n "C3 lazy: "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_lazy.so run
echo -n "C5 base: "; run --workload C5 --steps 100 --warmup 30
echo -n "C5M base: "; run --workload C5M --steps 100 --warmup 30
