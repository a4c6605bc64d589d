# wide order-log store in k_step_events: parity + same-box A/B against build_variants/lib_prev.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_device_ingress.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
FUZZ_LO=54000 FUZZ_HI=54300 python3 scripts/fuzz_host.py 2>&1 | tail -2
FUZZ_LO=4000 FUZZ_HI=4300 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -1
for rep in 1 2; do for lib in in-tree build_variants/lib_prev.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; for b in 8192 65536; do echo -n "$lib, $b books: "; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | grep -o "[0-9.]* M book-steps/s\|k_step_events: [0-9.]* ms" | tr "\n" " "; echo; done; done; done 2>&1 | tee $O/ab_log_store.txt
