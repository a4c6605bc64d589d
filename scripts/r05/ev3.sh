R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for v in A B C; do export BOURSE_AMD_LIBRARY=$R/build_variants/lib_evocc$v.so; for b in 8192 65536; do echo -n "occ variant $v, $b books: "; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | grep -o "[0-9.]* M book-steps/s\|k_step_events: [0-9.]* ms" | tr "\n" " "; echo; done; done; done 2>&1 | tee $O/ev_occ_variants.txt
export BOURSE_AMD_LIBRARY=$R/build_variants/lib_evoccB.so
python -m pytest tests/test_gpu_device_ingress.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
