# keyed k_step_events: register budgets (same box): in-tree (R = 4: 64 VGPRs + 44 B scratch, 8 waves; R = 8: 96 + 8 B, 5 waves),
# lib_evk_occ7 (R = 4: 69 VGPRs, 7 waves), lib_evk_occ6 (R = 8: 105 VGPRs, 4 waves), lib_prev (round 5 before the keyed form)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for lib in in-tree build_variants/lib_evk_occ7.so build_variants/lib_evk_occ6.so build_variants/lib_prev.so; do if [ $lib != in-tree ]; then export BOURSE_AMD_LIBRARY=$R/$lib; else unset BOURSE_AMD_LIBRARY; fi; for b in 8192 65536; do echo "== $lib, $b books"; python scripts/device_ingress_rate.py $b 2>&1 | grep -v amdgpu.ids; done; done; done 2>&1 | tee $O/ev_keyed_occ.txt
