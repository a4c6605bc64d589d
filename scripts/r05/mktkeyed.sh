# markets' step batches on the keyed / assembly loops (k_step_batch<R, MKT>): parity, soak, rates beside the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "market" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 900 python3 scripts/soak_agents.py 2>&1 | tail -2
for rep in 1 2; do python scripts/market_rate.py 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/market_rate_keyed.txt
