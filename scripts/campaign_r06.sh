# Round-6 validation campaign on the round's final code (GPU box): the fuzz drivers over fresh seed ranges + the soaks.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/campaign_r06
mkdir -p $O
FUZZ_LO=${LO_RANDOM:-800000} FUZZ_HI=${N_RANDOM:-803000} timeout 1500 python3 scripts/fuzz_random.py > $O/fuzz_random.txt 2>&1
FUZZ_LO=${LO_INGRESS:-20000} FUZZ_HI=${N_INGRESS:-21500} timeout 900 python3 scripts/fuzz_device_ingress.py > $O/fuzz_device_ingress.txt 2>&1
FUZZ_LO=${LO_HOST:-80000} FUZZ_HI=${N_HOST:-83000} timeout 1200 python3 scripts/fuzz_host.py > $O/fuzz_host.txt 2>&1
timeout 1500 python3 scripts/fuzz_wave_members.py ${LO_MEMBERS:-1200000} ${N_MEMBERS:-1500} > $O/fuzz_wave_members.txt 2>&1
timeout 900 python3 scripts/soak.py 2000 > $O/soak.txt 2>&1
timeout 900 python3 scripts/c5m_fullsize_parity.py > $O/c5m_fullsize.txt 2>&1
tail -n 3 $O/*.txt
FUZZ_LO=${LO_PARTS:-1300000} FUZZ_HI=${N_PARTS:-1300400} timeout 1500 python3 scripts/fuzz_parts.py > $O/fuzz_parts.txt 2>&1
timeout 900 python3 scripts/soak_agents.py > $O/soak_agents.txt 2>&1
tail -n 2 $O/fuzz_parts.txt $O/soak_agents.txt
# the host-driven step's keyed form - round 6: with modifications, market orders on every pool size and queues longer than the pool -
# every book against its oracle env (>= 20 000 configurations); RandomMarketAgents shapes against ManyMarkets
FUZZ_LO=${LO_KEYED:-100000} FUZZ_HI=${N_KEYED:-122000} timeout 2400 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee $O/fuzz_keyed_events.txt
FUZZ_LO=${LO_MARKETS:-10000} FUZZ_HI=${N_MARKETS:-11000} timeout 900 python3 scripts/fuzz_markets.py 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee $O/fuzz_markets.txt
