R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "market or host" 2>&1 | tail -2
FUZZ_LO=61000 FUZZ_HI=61300 python3 scripts/fuzz_host.py 2>&1 | tail -3
FUZZ_LO=9000 FUZZ_HI=9500 python3 scripts/fuzz_device_ingress.py 2>&1 | tail -1
for rep in 1 2; do for lib in in-tree build_variants/lib_head.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; for b in 8192 65536; do echo -n "$lib $b: "; python scripts/device_ingress_rate.py $b 2>&1 | grep -o "[0-9.]* M book-steps/s" | head -1; done; done; done | tee $O/ab_ev_markets3.txt
