# the campaign once more on the round's very last code (markets' lists on the keyed loops included)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
LO_RANDOM=700000 N_RANDOM=706000 N_INGRESS=1500 N_HOST=51200 LO_MEMBERS=990000 N_MEMBERS=2500 LO_PARTS=995000 N_PARTS=995400 bash scripts/campaign_r05.sh 2>&1 | grep -v amdgpu.ids | tail -36
FUZZ_LO=70000 FUZZ_HI=74000 timeout 900 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/campaign_r05/fuzz_keyed_events.txt
FUZZ_LO=5000 FUZZ_HI=8000 timeout 1500 python3 scripts/fuzz_markets.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/campaign_r05/fuzz_markets.txt
