# same-box timing of variant builds that leave a phase out (results are then wrong: timing only):  bash scripts/exp_variants.sh lib1.so [lib2.so ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f) ev/bs %.1f tr/bs %.1f %s' % (d['value']/1e6, d['runs']['median']/1e6, d['config']['events_per_book_step'], d['config']['trades_per_book_step'], {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for rep in 1 2; do
for lib in in-tree "$@"; do
  if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi
  echo "== $lib"
  for w in ${WORKLOADS:-C2 8192 C5}; do
    case $w in
      C3) echo -n "C3     "; run --steps 200 --warmup 50 ;;
      C5|C5M|C2) printf "%-7s" $w; run --workload $w --steps 100 --warmup 30 ;;
      *) printf "%-7s" $w; run --books $w --steps 200 --warmup 50 ;;
    esac
  done
done
done
