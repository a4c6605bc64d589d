# occupancy sensitivity of the two wave-per-book decodes (LDS padding lowers their waves per SIMD)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 100 --warmup 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for pad in 0 16384 49152; do echo "C5M  BOURSE_AMD_MW_LDS_PAD=$pad (2 x 75 KB workgroups per CU at 0 = 4 waves/SIMD; 1 workgroup = 2 waves/SIMD beyond ~5 KB)"; BOURSE_AMD_MW_LDS_PAD=$pad run --workload C5M; done
for pad in 0 4096 12288 24576; do echo "C5   BOURSE_AMD_WAVE_LDS_PAD=$pad (28.7 KB per 4-wave workgroup at 0 = 5 waves/SIMD; 4 / 3 / 3 beyond)"; BOURSE_AMD_WAVE_LDS_PAD=$pad run --workload C5; done
for pad in 0 8192 24576; do echo "shard BOURSE_AMD_WAVE_LDS_PAD=$pad (18.4 KB per workgroup at 0 = 8 waves/SIMD; 6 / 3 beyond)"; BOURSE_AMD_WAVE_LDS_PAD=$pad run --books 8192; done
