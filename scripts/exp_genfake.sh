R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f) ev/bs %.1f tr/bs %.1f %s' % (d['value']/1e6, d['runs']['median']/1e6, d['config']['events_per_book_step'], d['config']['trades_per_book_step'], {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for rep in 1 2; do
for lib in in-tree build_variants/libbourse_amd_genfake1.so build_variants/libbourse_amd_genfake2.so; do
  if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi
  echo "== $lib"
  echo -n "C2     "; run --workload C2 --steps 100 --warmup 30
  echo -n "8192   "; run --books 8192 --steps 200 --warmup 50
  echo -n "C5     "; run --workload C5 --steps 100 --warmup 30
done
done
