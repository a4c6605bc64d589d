# Where does k_agents_mixed_lanes (the C5-as-written members' update) spend its time?  SQ counters, GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { d=$1; shift; rocprofv3 --pmc "$@" -d $R/gpurun_out/$d -o p -f csv -- python3 $R/bench.py --workload C5M --steps 20 --warmup 10 --no-cpu-baseline --profile-every 0 --preheat-steps 0 --repeats 0 > /dev/null 2> $R/gpurun_out/$d.err; }
run pmc_c5m1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
run pmc_c5m2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES
run pmc_c5m3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVES
python3 - <<PY
import csv, glob, collections
for d in ("pmc_c5m1", "pmc_c5m2", "pmc_c5m3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-44:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "mixed_lanes" in k or "mixed_wave" in k or "k_step_batch" in k:
            print(d, k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
