R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
echo "== parity with BOURSE_AMD_STEP_DECODE=1"; BOURSE_AMD_STEP_DECODE=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
BOURSE_AMD_STEP_DECODE=1 FUZZ_LO=600000 FUZZ_HI=600300 python scripts/fuzz_random.py 2>&1 | tail -2
run() { python3 bench.py --no-cpu-baseline --repeats 2 --steps 200 --warmup 50 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('%8.1f M (median %8.1f)  %s' % (d['value']/1e6, d['runs']['median']/1e6, {a: round(b['avg_launch_ms']*1e3,1) for a,b in k.items()}))"; }
for rep in 1 2; do for sd in 0 1; do export BOURSE_AMD_STEP_DECODE=$sd; for b in 8192 12288 16384; do echo -n "STEP_DECODE=$sd books $b: "; run --books $b; done; echo -n "STEP_DECODE=$sd C5: "; run --workload C5 --steps 100 --warmup 30; done; done 2>&1 | tee $O/ab_step_decode.txt
