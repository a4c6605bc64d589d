R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "market or host" 2>&1 | tail -2
FUZZ_LO=62000 FUZZ_HI=62400 python3 scripts/fuzz_host.py 2>&1 | tail -3
BOURSE_AMD_EV_WAVE_SHUFFLE_MIN=2 FUZZ_LO=62400 FUZZ_HI=62800 python3 scripts/fuzz_host.py 2>&1 | tail -3
FUZZ_LO=9500 FUZZ_HI=10000 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -1
for rep in 1 2; do for seq in 0 1; do export BOURSE_AMD_EV_SEQ_SHUFFLE=$seq; for cfg in "4096 2" "2048 4"; do echo -n "sequential shuffle = $seq: "; python scripts/market_ingress_rate.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130; done; done; done | tee $O/market_wave_shuffle.txt
