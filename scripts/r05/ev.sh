R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -3
for lib in in-tree build_variants/lib_prev.so in-tree build_variants/lib_prev.so; do
  if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi
  echo "== $lib"
  for b in 8192 16384 65536; do python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-330; done
done 2>&1 | tee $O/ab_step_events_occ.txt
