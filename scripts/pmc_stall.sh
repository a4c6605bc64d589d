# Where do k_step_batch's cycles go?  SQ activity / wait / instruction-cache counters, one part per launch (GPU box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export BOURSE_AMD_SPLIT_PARTS=1
run() { d=$1; shift; rocprofv3 --pmc "$@" -d $R/gpurun_out/$d -o p -f csv -- python3 $R/bench.py --steps 20 --warmup 20 --steps-per-launch 20 --no-cpu-baseline --profile-every 0 > /dev/null 2> $R/gpurun_out/$d.err; }
run pmc_stall1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_IFETCH
run pmc_stall2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_WAVES
python3 - <<PY
import csv, glob, collections
for d in ("pmc_stall1", "pmc_stall2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "k_step_batch" in k or "k_agents_fsm" in k:
            print(d, k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
tail -n 3 $R/gpurun_out/pmc_stall1.err; tail -n 3 $R/gpurun_out/pmc_stall2.err
