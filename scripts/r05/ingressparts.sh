R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python scripts/ingress_parts_probe.py 8192 2>&1 | grep -v amdgpu.ids | tee $O/ingress_parts_probe.txt
python scripts/ingress_parts_probe.py 65536 2>&1 | grep -v amdgpu.ids | tee -a $O/ingress_parts_probe.txt
