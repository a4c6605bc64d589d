R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -3
FUZZ_LO=3500 FUZZ_HI=3560 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -2
FUZZ_LO=6000 FUZZ_HI=9000 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -2
