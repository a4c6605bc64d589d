# what the sequential shuffle costs the keyed k_step_events: a build that leaves its swaps out (results then wrong, time only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for lib in in-tree build_variants/lib_evk_skip1.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; for b in 8192 65536; do echo "== $lib, $b books"; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-330; done; done; done 2>&1 | tee $O/ev_keyed_shuffle_share.txt
