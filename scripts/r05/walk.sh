R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
bash scripts/r05/occ.sh 2>&1 | tee $O/occupancy_probe.txt
echo "== parity on lib_walk"; BOURSE_AMD_LIBRARY=$R/build_variants/lib_walk.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
BOURSE_AMD_LIBRARY=$R/build_variants/lib_walk.so FUZZ_HI=150 python scripts/fuzz_random.py 2>&1 | tail -3
WORKLOADS="C5" bash scripts/exp_ab.sh build_variants/lib_walk.so 2>&1 | tee $O/ab_walk.txt
for b in 4096 16384; do for lib in in-tree build_variants/lib_walk.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; echo "256-slot pool, $lib:"; python3 scripts/shape_sweep.py R4 $b wave_split,wave 2>/dev/null | tail -1; done; done
