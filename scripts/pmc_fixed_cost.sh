# Scalar / vector instructions of k_step_batch per book-step as a function of the activity rate (0 = no events: the
# fixed load + snapshot + store part).  GPU box.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export BOURSE_AMD_SPLIT_PARTS=1
cat > /tmp/rate_run.py <<PY
import sys
sys.path.insert(0, "$R")
import bourse_amd
rate = float(sys.argv[1]); B = 65536
env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=32, max_live_orders=128, trade_capacity=4096, history_capacity=8)
env.set_random_agents([(64, (32, 64), (10, 20), 2, rate), (64, (32, 64), (50, 70), 2, rate)])
env.set_pipeline("split")
env.run(30); env.sync()
print(rate, int(env.trade_counts().sum()) / (B * 30.0))
PY
for rate in 0.0 0.25 0.5 1.0; do
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_WAVES -d $R/gpurun_out/pmc_rate_$rate -o p -f csv -- python3 /tmp/rate_run.py $rate 2> /dev/null | tail -n 1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_rate_$rate/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-36:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "k_step_batch" in k:
        # steady state: the last 10 launches
        print("  rate $rate", k, {c: round(sum(x[-10:]) / 10 / 65536) for c, x in v.items()})
PY
  rm -rf $R/gpurun_out/pmc_rate_$rate
done
