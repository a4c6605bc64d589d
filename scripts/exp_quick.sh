# quick A/B of the in-tree library on the headline and the shard (GPU box): bash scripts/exp_quick.sh [label]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
echo "== ${1:-in-tree}"
echo -n "C3        "; run
echo -n "C3 again  "; run
echo -n "C3 20/5   "; run --steps 20 --warmup 5
echo -n "8192      "; run --books 8192
echo -n "16384     "; run --books 16384
echo -n "32768     "; run --books 32768
echo -n "C2        "; run --workload C2
echo -n "C5        "; run --workload C5 --steps 100 --warmup 30
echo -n "C5M       "; run --workload C5M --steps 100 --warmup 30
