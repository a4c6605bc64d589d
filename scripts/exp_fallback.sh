R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 1 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M  %s' % (d['value']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for W in C5M C5; do
echo -n "$W keyed (in-tree):        "; run --workload $W
echo -n "$W two-reduction loop only: "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_nokey.so run --workload $W
done
echo -n "C3 keyed:  "; run --steps 200 --warmup 50
echo -n "C3 two-reduction only: "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_nokey.so run --steps 200 --warmup 50
