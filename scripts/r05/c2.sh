# C2 (4 096 books x 64 agents x 16 levels): where the time of a book-step goes, and where the knee is
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05
for b in 4096 2048; do BOURSE_AMD_LIBRARY=$R/build_variants/lib_stamps.so BOURSE_AMD_WAVE_PARTS=1 python scripts/wave_phases.py $b wave_split C2 2>&1 | grep -v amdgpu.ids; done
python scripts/shape_sweep.py C2 1024,2048,3072,4096,5120,6144,8192,12288 wave_split,wave 2>/dev/null
