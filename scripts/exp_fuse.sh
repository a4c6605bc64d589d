# k_step_decode (events of step s + decode of step s + 1 in one launch): on / off, occupancy variants.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s  %s' % (d['value']/1e6, d['runs']['median']/1e6, d['config']['pipeline'][:32], {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for B in 8192 12288 16384 24000; do
  echo -n "books $B separate launches: "; BOURSE_AMD_FUSE_STEPS=0 run --books $B --pipeline wave_split
  echo -n "books $B fused, occ 7:      "; run --books $B --pipeline wave_split
  echo -n "books $B fused, occ 6:      "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_sd6.so run --books $B --pipeline wave_split
  echo -n "books $B fused, occ 8:      "; BOURSE_AMD_LIBRARY=$R/build_variants/lib_sd8.so run --books $B --pipeline wave_split
  for P in 2 4; do echo -n "books $B fused, occ 7, $P parts: "; run --books $B --pipeline wave_split --wave-parts $P; done
done
