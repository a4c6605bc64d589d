R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_statistics.py -m gpu -x -q 2>&1 | tail -3
python scripts/fuzz_wave_members.py 860000 1500 2>&1 | tail -2
python scripts/c5m_fullsize_parity.py 2>&1 | tail -1
WORKLOADS="C5M C5" bash scripts/exp_ab.sh build_variants/lib_nowide.so 2>&1 | tee $O/ab_keyed_wide.txt
