# SQ instruction counters per kernel of one bench workload (default C5): bash scripts/pmc_workload.sh [workload].  GPU box.
W=${1:-C5}
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
run() { d=$1; shift; rocprofv3 --pmc "$@" -d $R/gpurun_out/$d -o p -f csv -- python3 $R/bench.py --workload $W --steps 20 --warmup 10 --no-cpu-baseline --profile-every 0 --preheat-steps 0 --repeats 0 > /dev/null 2> $R/gpurun_out/$d.err; }
run pmc_w1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
run pmc_w2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES
python3 - <<PY
import csv, glob, collections
for d in ("pmc_w1", "pmc_w2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-44:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "k_agents" in k or "k_step_batch" in k or "k_run" in k:
            w = v["SQ_WAVES"] / n[(k, "SQ_WAVES")]
            print("$W", d, k, "waves/launch", round(w), {c: round(x / n[(k, c)] / w) for c, x in v.items() if c != "SQ_WAVES"}, "(per wave)")
PY
rm -rf $R/gpurun_out/pmc_w1 $R/gpurun_out/pmc_w2
