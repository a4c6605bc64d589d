R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/prof_r05; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --workload C5M --steps 100 --warmup 30 > $O/bench_C5M.json 2> $O/bench_C5M.err
python3 bench.py --workload C5 --steps 100 --warmup 30 > $O/bench_C5.json 2> $O/bench_C5.err
bash scripts/pmc_all.sh r05 C5M:8192 2>&1 | tail -6
python3 -c "
import json
for w in ('C5M','C5'):
    d=json.loads([l for l in open('$O/bench_%s.json' % w) if l.startswith('{')][-1]); r=d['roofline']
    print(w, round(d['value']/1e6,1), 'median', round(d['runs']['median']/1e6,1), 'frac %.3f node %.3f standalone %.3f' % (r['frac'], r['frac_node'], r.get('standalone',{}).get('frac',0)), {k: round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()}, 'cpu', round(d['cpu_baseline']['value']/1e6,2))
"
