R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_bench_options.py -m gpu -x -q 2>&1 | tail -12
python bench.py --workload INGRESS --steps 24 --warmup 6 2>/dev/null | tail -1 > $O/bench_INGRESS.json; cat $O/bench_INGRESS.json | cut -c1-1500
python bench.py --workload INGRESS --books 65536 --steps 24 --warmup 6 2>/dev/null | tail -1 > $O/bench_INGRESS_65536.json; cat $O/bench_INGRESS_65536.json | cut -c1-600
