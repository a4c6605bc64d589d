#!/bin/bash
# Bench every compiler-flag variant in build_variants/*.so on the C3 workload (GPU box; the product .so is restored at the end).
LIB=bourse_amd/csrc/libbourse_amd.so
cp $LIB /tmp/keep.so
for v in build_variants/*.so; do
  cp $v $LIB
  printf "%-12s " $(basename $v .so)
  timeout 300 python bench.py --steps 150 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('%.1f M  K_A %.1f us  K_B %.1f us' % (d['value']/1e6, k['k_agents_fsm']['avg_launch_ms']*1e3, k['k_step_batch']['avg_launch_ms']*1e3))"
done
cp /tmp/keep.so $LIB
