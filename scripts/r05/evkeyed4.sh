R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
FUZZ_LO=0 FUZZ_HI=1500 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/fuzz_keyed_events.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for b in 8192 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids; done | tee $O/device_ingress_rate.txt
