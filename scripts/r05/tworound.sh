R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_statistics.py -m gpu -x -q 2>&1 | tail -3
python scripts/fuzz_wave_members.py 890000 2000 2>&1 | tail -2
python scripts/c5m_fullsize_parity.py 2>&1 | tail -1
python scripts/soak_agents.py 2>&1 | tail -2
WORKLOADS="C5M" bash scripts/exp_ab.sh build_variants/lib_buckets.so 2>&1 | tee $O/ab_two_round.txt
