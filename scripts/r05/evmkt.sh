R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "market" 2>&1 | tail -3
FUZZ_LO=60000 FUZZ_HI=60400 python3 scripts/fuzz_host.py 2>&1 | tail -3
FUZZ_LO=8000 FUZZ_HI=8600 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -1
python scripts/market_rate.py 2>&1 | grep -v amdgpu.ids | tail -4
