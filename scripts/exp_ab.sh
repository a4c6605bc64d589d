# same-box A/B of the in-tree library against variant libraries on the headline, the shard and C5 (GPU box):
#   bash scripts/exp_ab.sh build_var/lib_prev.so [...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for rep in 1 2; do
for lib in in-tree "$@"; do
  echo "== $lib"
  if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi
  echo -n "C3        "; run
  echo -n "8192      "; run --books 8192
  echo -n "C5        "; run --workload C5 --steps 100 --warmup 30
done
done
