R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
WORKLOADS="C3 8192 16384 C2" bash scripts/exp_ab.sh build_variants/lib_prev.so 2>&1 | tee $O/ab_sload.txt
BOURSE_AMD_LIBRARY=$R/build_variants/lib_stamps.so python scripts/wave_phases.py 8192 auto C3 --skew 2>&1 | tee $O/wave_phases_8192.txt
BOURSE_AMD_LIBRARY=$R/build_variants/lib_stamps.so python scripts/wave_phases.py 16384 auto C3 2>&1 | tee $O/wave_phases_16384.txt
