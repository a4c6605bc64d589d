R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
FUZZ_LO=52000 FUZZ_HI=52400 python3 scripts/fuzz_host.py 2>&1 | tail -3
FUZZ_LO=2000 FUZZ_HI=2400 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -1
for b in 8192 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-330; done | tee $O/ingress_vec_shuffle.txt
bash scripts/pmc_events.sh 8192 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/host_driven_rate.py 8192 2>&1 | grep -v amdgpu.ids | cut -c1-200
