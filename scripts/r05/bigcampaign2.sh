# the round's last big campaign (after the keyed host-driven step and the wave-parallel shuffle in k_step_events): fresh seeds
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
LO_RANDOM=640000 N_RANDOM=660000 N_INGRESS=4000 N_HOST=54500 LO_MEMBERS=940000 N_MEMBERS=6000 LO_PARTS=990000 N_PARTS=991000 bash scripts/campaign_r05.sh 2>&1 | grep -v amdgpu.ids | tail -40
FUZZ_LO=30000 FUZZ_HI=42000 timeout 1500 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/campaign_r05/fuzz_keyed_events.txt
BOURSE_AMD_EV_WAVE_SHUFFLE_MIN=2 FUZZ_LO=42000 FUZZ_HI=48000 timeout 1200 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/campaign_r05/fuzz_keyed_events_wave_shuffle.txt
