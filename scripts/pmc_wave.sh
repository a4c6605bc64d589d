# k_agents_wave / k_step_batch at B books (default 8192): kernel times (rocprofv3 --kernel-trace --stats) and SQ counters.
# GPU box:  bash scripts/pmc_wave.sh [books] [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8192}
TAG=${2:-wave}
PIPE=${3:-wave}
ARGS="--books $B --steps 40 --warmup 20 --steps-per-launch 20 --no-cpu-baseline --profile-every 0 --repeats 0 --pipeline $PIPE"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$TAG -o kt -f csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/kt_$TAG.json 2> $R/gpurun_out/kt_$TAG.err
run() { d=$1; shift; rocprofv3 --pmc "$@" -d $R/gpurun_out/$d -o p -f csv -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/$d.err; }
run pmc_${TAG}1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVES
run pmc_${TAG}2 SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/kt_$TAG/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:60], r["Calls"], "avg_ns", r["AverageNs"], "pct", r["Percentage"])
for d in ("pmc_${TAG}1", "pmc_${TAG}2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "k_step_batch" in k or "k_agents" in k:
            print(d, k, {c: round(x / n[(k, c)] / $B, 1) for c, x in v.items()}, "(per book-step)")
PY
tail -n 2 $R/gpurun_out/kt_$TAG.err
