# wave-parallel shuffle in k_step_events: parity, then the same library with BOURSE_AMD_EV_SEQ_SHUFFLE=1 (the draw-by-draw loop) beside it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25
FUZZ_LO=2000 FUZZ_HI=2600 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -6
FUZZ_LO=56000 FUZZ_HI=56300 python3 scripts/fuzz_host.py 2>&1 | tail -4
for rep in 1 2; do for seq in 0 1; do export BOURSE_AMD_EV_SEQ_SHUFFLE=$seq; for b in 8192 65536; do echo "== sequential shuffle = $seq, $b books"; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-330; done; done; done 2>&1 | tee $O/ab_ev_wave_shuffle.txt
