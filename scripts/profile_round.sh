# Round profile: rocprofv3 kernel stats of the bench commands + HBM traffic / instruction counters (separate passes).
# GPU box:  bash scripts/profile_round.sh r02     -> gpurun_out/prof_r02/ (copy the summaries to profiles/r02/)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# bench lines (un-profiled): headline C3, its 8-GPU shard (C4: 8 192 books per GPU), the other BASELINE configs
python3 $R/bench.py --steps 200 --warmup 50 > $OUT/bench_C3.json 2> $OUT/bench_C3.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C3_driver_args.json 2>> $OUT/bench_C3.err
python3 $R/bench.py --books 8192 --steps 200 --warmup 50 --no-cpu-baseline > $OUT/bench_C4_shard_8192.json 2> $OUT/bench_C4.err
for N in 32768 16384; do python3 $R/bench.py --books $N --steps 100 --warmup 50 --no-cpu-baseline > $OUT/bench_C4_shard_$N.json 2>> $OUT/bench_C4.err; done
for W in C2 C5 C5M; do python3 $R/bench.py --workload $W --steps 100 --warmup 30 > $OUT/bench_$W.json 2> $OUT/bench_$W.err; done
# kernel stats (no pre-heat env here: its launches would be averaged into the same kernel names)
rocprofv3 --kernel-trace --stats -d $OUT/kt_C3 -o kt -f csv -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --preheat-steps 0 > $OUT/bench_C3_under_rocprof.json 2> $OUT/kt_C3.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_C4 -o kt -f csv -- python3 $R/bench.py --books 8192 --steps 200 --warmup 50 --no-cpu-baseline --preheat-steps 0 > $OUT/bench_C4_under_rocprof.json 2> $OUT/kt_C4.err
# PMC passes (counters only; FETCH_SIZE and WRITE_SIZE cannot share a pass)
PA="--steps 50 --warmup 50 --no-cpu-baseline --profile-every 0 --repeats 0"
for B in 65536 8192; do
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$B -o p -f csv -- python3 $R/bench.py --books $B $PA > /dev/null 2> $OUT/pmc_fetch_$B.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$B -o p -f csv -- python3 $R/bench.py --books $B $PA > /dev/null 2> $OUT/pmc_write_$B.err
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_sq_$B -o p -f csv -- python3 $R/bench.py --books $B $PA > /dev/null 2> $OUT/pmc_sq_$B.err
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for B in (65536, 8192):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for kind in ("fetch", "write", "sq"):
        for f in glob.glob("$OUT/pmc_%s_%d/**/*counter_collection.csv" % (kind, B), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    out[B] = {k: {c: v[c] / n[(k, c)] for c in v} for k, v in agg.items() if k.startswith("k_")}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for B, d in out.items():
    for k, v in d.items():
        print(B, k, {c: round(x, 1) for c, x in v.items()})
PY
find $OUT -name "*kernel_stats.csv" | head; tail -n 2 $OUT/*.err | tail -20
