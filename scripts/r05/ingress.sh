R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -15
python scripts/host_driven_rate.py 8192 2>&1 | tee $O/host_driven_rate.txt
python scripts/host_driven_rate.py 65536 2>&1 | tee -a $O/host_driven_rate.txt
