R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver-args bench: %.1f M, runs %s, roofline frac %.3f, cpu %.2f M' % (d['value']/1e6, [round(v/1e6,1) for v in d['runs']['values']], d['roofline']['frac'], d['cpu_baseline']['value']/1e6))"
