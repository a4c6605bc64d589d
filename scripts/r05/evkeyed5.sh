R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for b in 8192 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids; done | tee $O/device_ingress_rate.txt
for b in 8192 65536; do python scripts/host_driven_rate.py $b 2>&1 | grep -v amdgpu.ids; done | tee $O/host_driven_rate.txt
bash scripts/pmc_events.sh 2>&1 | tail -30 | tee $O/pmc_step_events_keyed.txt
