# Round-5 validation campaign on the round's final code (GPU box): the fuzz drivers over fresh seed ranges + the soaks.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/campaign_r05
mkdir -p $O
FUZZ_LO=${LO_RANDOM:-400000} FUZZ_HI=${N_RANDOM:-403000} timeout 1500 python3 scripts/fuzz_random.py > $O/fuzz_random.txt 2>&1
FUZZ_LO=0 FUZZ_HI=${N_INGRESS:-600} timeout 900 python3 scripts/fuzz_device_ingress.py > $O/fuzz_device_ingress.txt 2>&1
FUZZ_LO=50000 FUZZ_HI=${N_HOST:-50600} timeout 1200 python3 scripts/fuzz_host.py > $O/fuzz_host.txt 2>&1
timeout 1500 python3 scripts/fuzz_wave_members.py ${LO_MEMBERS:-700000} ${N_MEMBERS:-1500} > $O/fuzz_wave_members.txt 2>&1
timeout 900 python3 scripts/soak.py 2000 > $O/soak.txt 2>&1
timeout 900 python3 scripts/c5m_fullsize_parity.py > $O/c5m_fullsize.txt 2>&1
tail -n 3 $O/*.txt
FUZZ_LO=${LO_PARTS:-900000} FUZZ_HI=${N_PARTS:-900400} timeout 1500 python3 scripts/fuzz_parts.py > $O/fuzz_parts.txt 2>&1
timeout 900 python3 scripts/soak_agents.py > $O/soak_agents.txt 2>&1
tail -n 2 $O/fuzz_parts.txt $O/soak_agents.txt
# (late round 5) the host-driven step's keyed form, every book against its oracle env; RandomMarketAgents shapes against ManyMarkets
FUZZ_LO=${LO_KEYED:-0} FUZZ_HI=${N_KEYED:-1500} timeout 900 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee $O/fuzz_keyed_events.txt
FUZZ_LO=${LO_MARKETS:-0} FUZZ_HI=${N_MARKETS:-1000} timeout 900 python3 scripts/fuzz_markets.py 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee $O/fuzz_markets.txt
