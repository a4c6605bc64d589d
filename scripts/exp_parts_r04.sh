# wave_split in 2 / 3 / 4 parts at the C4 shard sizes, on the round's final kernels (GPU box): bash scripts/exp_parts_r04.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for B in 8192 12288 16384; do for p in 2 3 4; do echo -n "$B parts $p: "; run --books $B --pipeline wave_split --wave-parts $p; done; done
