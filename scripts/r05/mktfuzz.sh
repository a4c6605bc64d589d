R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
FUZZ_LO=2000 FUZZ_HI=4000 timeout 1500 python3 scripts/fuzz_markets.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/fuzz_markets.txt
