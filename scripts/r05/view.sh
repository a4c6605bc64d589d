R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -2
python scripts/host_driven_rate.py 8192 2>&1 | grep -v amdgpu.ids | cut -c1-200
python scripts/host_driven_rate.py 65536 2>&1 | grep -v amdgpu.ids | cut -c1-200
