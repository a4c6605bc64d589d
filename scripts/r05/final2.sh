# campaign on the round's last code (after the keyed host-driven step): every fuzz driver over fresh seeds + soaks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
LO_RANDOM=430000 N_RANDOM=433000 N_INGRESS=1200 N_HOST=51500 LO_MEMBERS=730000 N_MEMBERS=1500 LO_PARTS=930000 N_PARTS=930400 bash scripts/campaign_r05.sh 2>&1 | tail -40
FUZZ_LO=10000 FUZZ_HI=13000 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/campaign_r05/fuzz_keyed_events.txt
