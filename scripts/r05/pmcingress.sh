R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash scripts/pmc_all.sh ingress INGRESS:8192 INGRESS:65536 2>&1 | tail -12
