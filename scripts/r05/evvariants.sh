# the host-driven step's keyed form on the narrowed key windows (steps alternate between the keyed and the event-by-event loop
# because of the WINDOW, not the events): lib_sb8 = 8-bit arrival field, lib_sb24 = 7-bit price field
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for f in build_variants/lib_sb8.so build_variants/lib_sb24.so; do
  echo "== $f"
  BOURSE_AMD_LIBRARY=$R/$f timeout 600 python -m pytest tests/test_gpu_keyed_events.py tests/test_gpu_device_ingress.py -m gpu -q 2>&1 | tail -8
  BOURSE_AMD_LIBRARY=$R/$f FUZZ_LO=20000 FUZZ_HI=20800 timeout 600 python3 scripts/fuzz_keyed_events.py 2>&1 | grep -v amdgpu.ids | tail -4
  BOURSE_AMD_LIBRARY=$R/$f FUZZ_LO=58000 FUZZ_HI=58200 timeout 600 python3 scripts/fuzz_host.py 2>&1 | tail -3
done 2>&1 | tee $O/ev_keyed_variants.txt
