R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for pad in 0 16384; do echo -n "C5M  BOURSE_AMD_MW_LDS_PAD=$pad: "; BOURSE_AMD_MW_LDS_PAD=$pad run --workload C5M; done
for p in 2 3 4 6 8; do echo -n "C5M wave parts $p: "; run --workload C5M --wave-parts $p; done
for b in 2048 4096 6144 12288 16384; do echo -n "C5M shape, $b books: "; run --workload C5M --books $b; done
