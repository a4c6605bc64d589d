R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
LO_RANDOM=500000 N_RANDOM=501000 N_INGRESS=400 N_HOST=50300 LO_MEMBERS=800000 N_MEMBERS=500 LO_PARTS=950000 N_PARTS=950150 bash scripts/campaign_r05.sh 2>&1 | grep -v amdgpu.ids | tail -40
