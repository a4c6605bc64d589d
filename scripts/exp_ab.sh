# same-box A/B of the in-tree library against variant libraries (GPU box):
#   [WORKLOADS="C3 8192 C5 C5M"] bash scripts/exp_ab.sh build_var/lib_prev.so [...]
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
  for w in ${WORKLOADS:-C3 8192 C5 C5M}; do
    case $w in
      C3) echo -n "C3        "; run ;;
      C5|C5M|C2) printf "%-10s" $w; run --workload $w --steps 100 --warmup 30 ;;
      *) printf "%-10s" $w; run --books $w ;;
    esac
  done
done
done
