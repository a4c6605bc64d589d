# same-box A/B of the external-agents stream (bench.py --workload INGRESS) between the in-tree library and variant builds:
#   bash scripts/ingress_ab.sh build_variants/libX.so [...]     clean and mixed stream, 8 192 and 65 536 books, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --workload INGRESS --steps 24 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%6.1f M  keyed %.4f  k_step_events %.1f us' % (d['value']/1e6, d['keyed_frac'], d['roofline']['avg_launch_ms']*1e3))"; }
for rep in 1 2; do
for lib in in-tree "$@"; do
  if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi
  echo "== $lib"
  echo -n "   8192 clean  "; run
  echo -n "   8192 mixed  "; run --modify-frac 0.05 --market-frac 0.02
  echo -n "  65536 clean  "; run --books 65536
  echo -n "  65536 mixed  "; run --books 65536 --modify-frac 0.05 --market-frac 0.02
done
done
