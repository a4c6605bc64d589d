# where a C5 wave's time goes under the real pipeline's load (stamps build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
BOURSE_AMD_LIBRARY=$R/build_variants/lib_stamps.so python scripts/wave_phases.py 8192 auto C5 2>&1 | grep -v amdgpu.ids | tee $O/wave_phases_C5.txt
