# wave_split parts at the C5 workloads (the library's defaults: 3 for RandomAgents, 2 for Noise / Momentum sets).  GPU box.
for w in C5M C5; do for p in 1 2 3 4; do
  python bench.py --workload $w --wave-parts $p --no-cpu-baseline --repeats 2 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$w parts $p', round(j['value'] / 1e6, 2), round(j['runs']['median'] / 1e6, 2), {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in j['roofline']['kernels'].items()})"
done; done
