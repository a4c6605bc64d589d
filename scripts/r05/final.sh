# Final pass of round 5 on the round's last code: GPU suite, smoke, the full profile, a larger campaign
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile_round5.sh r05 2>&1 | grep -v amdgpu.ids | tail -45
LO_RANDOM=520000 N_RANDOM=524000 N_INGRESS=1500 N_HOST=51200 LO_MEMBERS=820000 N_MEMBERS=2500 LO_PARTS=960000 N_PARTS=960400 bash scripts/campaign_r05.sh 2>&1 | grep -v amdgpu.ids | tail -40
