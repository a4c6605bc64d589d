# k_step_events on the keyed loop: parity + same-box A/B against build_variants/lib_prev.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_device_ingress.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25
FUZZ_LO=55000 FUZZ_HI=55400 python3 scripts/fuzz_host.py 2>&1 | tail -4
FUZZ_LO=5000 FUZZ_HI=5400 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -3
for rep in 1 2; do for lib in in-tree build_variants/lib_prev.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; for b in 8192 65536; do echo -n "$lib, $b books: "; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | grep -o "[0-9.]* M book-steps/s\|k_step_events: [0-9.]* ms" | tr "\n" " "; echo; done; done; done 2>&1 | tee $O/ab_ev_keyed.txt
