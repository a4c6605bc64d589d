# members' market lists (MarketAgentSet) on the keyed loops: parity + fuzz + same-box A/B against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_statistics.py -m gpu -x -q -k "market or member or agent_set or fuzz" 2>&1 | tail -3
timeout 900 python3 scripts/fuzz_many.py 2>&1 | tail -2
FUZZ_LO=996000 FUZZ_HI=996300 timeout 1200 python3 scripts/fuzz_parts.py 2>&1 | tail -2
timeout 900 python3 scripts/soak_agents.py 2>&1 | tail -2
for rep in 1 2; do for lib in in-tree build_variants/lib_head.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; echo "== $lib"; python scripts/market_members_rate.py 2>&1 | grep -v amdgpu.ids | tail -2; done; done | tee $O/market_members_rate.txt
