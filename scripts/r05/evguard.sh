R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -2
FUZZ_LO=50000 FUZZ_HI=51500 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2; do for b in 8192 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-200; done; done | tee $O/device_ingress_rate_guards.txt
