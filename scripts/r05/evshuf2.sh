# wave-parallel shuffle in k_step_events: where it pays (queue length x pool), same library, BOURSE_AMD_EV_SEQ_SHUFFLE / _MIN
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -3
export BOURSE_AMD_EV_WAVE_SHUFFLE_MIN=2
for cfg in "8192 16 64" "8192 32 128" "8192 24 256" "8192 48 256" "8192 96 512" "8192 160 512" "65536 48 512" "32768 96 512"; do for seq in 0 1; do export BOURSE_AMD_EV_SEQ_SHUFFLE=$seq; echo -n "books instructions pool = $cfg, sequential shuffle = $seq: "; python scripts/device_ingress_rate.py $cfg 2>&1 | grep -o "[0-9.]* M book-steps/s\|k_step_events: [0-9.]* ms\|AssertionError.*" | tr "\n" " "; echo; done; done 2>&1 | tee $O/ev_wave_shuffle_sweep.txt
