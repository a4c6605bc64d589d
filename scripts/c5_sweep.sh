for v in "auto 3" "fused 3" "split 1" "split 2"; do set -- $v
echo "== pipeline=$1 parts=$2"
BOURSE_AMD_SPLIT_PARTS=$2 timeout 300 python bench.py --workload C5 --pipeline $1 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['ms_per_step'], d['config'].get('pipeline'), {k:v['avg_launch_ms'] for k,v in d['roofline']['kernels'].items()})"
done
