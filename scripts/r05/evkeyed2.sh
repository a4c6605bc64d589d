R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_keyed_events.py -m gpu -x -q 2>&1 | tail -40
