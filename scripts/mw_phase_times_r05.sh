# C5 as written / C5 stand-in: what the shuffle's RESOLUTION costs (variant build that returns after the draws' acceptance; results
# then wrong, the events are processed unshuffled).  GPU box.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
run() { python3 bench.py --no-cpu-baseline --repeats 1 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M  %s' % (d['value']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for lib in in-tree build_variants/lib_nores.so; do if [ $lib = in-tree ]; then unset BOURSE_AMD_LIBRARY; else export BOURSE_AMD_LIBRARY=$R/$lib; fi; for w in C5M C5; do echo -n "$lib $w: "; run --workload $w; done; echo -n "$lib shard: "; run --books 8192; done
