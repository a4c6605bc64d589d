# the contract region on fresh boxes: first region vs the median of five, what the pre-heat ran
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('value %.1f M, runs %s, preheat steps %s, chunk rates %s' % (d['value']/1e6, [round(v/1e6,1) for v in d['runs']['values']], d.get('preheat_steps'), [round(x/1e6,1) for x in c.get('preheat_chunk_rates', [])]))"; done 2>&1 | tee $O/bench_ramp_probe2.txt
python -m pytest tests/test_bench_options.py -m gpu -x -q 2>&1 | tail -2
