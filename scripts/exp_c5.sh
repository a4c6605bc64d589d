R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for LIB in "" lib_sb8_0.so lib_sb8_79.so lib_sb8_127.so; do
  for W in C5 C5M; do echo -n "$W ${LIB:-in-tree(claim 96)}: "; BOURSE_AMD_LIBRARY=${LIB:+$R/build_variants/$LIB} run --workload $W; done
done
for P in 2 4; do for W in C5 C5M; do echo -n "$W in-tree, $P parts: "; run --workload $W --wave-parts $P; done; done
