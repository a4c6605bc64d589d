# host arrays: staging copies and uploads in three overlapped groups, beside BOURSE_AMD_HI_ONE_PASS=1 (stage everything, then upload)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_device_ingress.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for one in 0 1; do export BOURSE_AMD_HI_ONE_PASS=$one; for b in 8192 65536; do echo "== one pass = $one, $b books"; python scripts/host_driven_rate.py $b 2>&1 | grep "^sync\|^tickets\|^views"; done; done; done 2>&1 | tee $O/ab_hi_groups.txt
