# Instruction-cache behaviour of the step kernels (round 5): SQC_ICACHE_* and the instruction-fetch level per kernel.
# k_step_batch<2> is ~6.2 k instructions (~40 KB of code: eight specialised copies of the New path) on a 64 KB instruction
# cache shared by the CUs of a group - is a wave's "waiting" partly instruction fetch?
# GPU box:  bash scripts/pmc_icache.sh [configs...]   -> gpurun_out/pmc_icache/summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
CONFIGS=${*:-"C3:65536 C3:8192 C5:8192"}
OUT=$R/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PA="--steps 40 --warmup 20 --steps-per-launch 20 --no-cpu-baseline --profile-every 0 --repeats 0 --preheat-steps 0"
for C in $CONFIGS; do
  W=${C%%:*}; B=${C##*:}
  run() { d=$1; shift; rocprofv3 --pmc "$@" -d $OUT/${W}_${B}_$d -o p -f csv -- python3 $R/bench.py --workload $W --books $B $PA > $OUT/${W}_${B}_$d.json 2> $OUT/${W}_${B}_$d.err
          grep -q '^{' $OUT/${W}_${B}_$d.json || echo "pmc_icache: $W:$B pass $d printed no bench line" >&2; }
  run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
  run if SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU
done
python3 - <<PY | tee $OUT/summary.txt
import collections, csv, glob, os, re
out = "$OUT"
for d in sorted(glob.glob(os.path.join(out, "*_ic"))):
    key = os.path.basename(d)[:-3]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); bs = collections.defaultdict(float)
    for kind in ("ic", "if"):
        seen = collections.defaultdict(float)
        for f in glob.glob(os.path.join(out, key + "_" + kind, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
                if not k.startswith("k_") or k in ("k_delay", "k_gather_header", "k_book_service", "k_stats", "k_flags_summary"):
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] in ("SQC_ICACHE_REQ", "SQ_IFETCH"):
                    g = float(r["Grid_Size"])
                    seen[k] += g if k in ("k_agents_fsm", "k_agents_mixed_lanes") else g / 64.0 * (20 if k.startswith("k_run") else 1)
        for k, v in seen.items():
            bs[(k, kind)] = v
    print(key)
    for k, c in agg.items():
        b1, b2 = bs[(k, "ic")] or 1, bs[(k, "if")] or 1
        req, hit, miss, dup = (c.get("SQC_ICACHE_" + n, 0) / b1 for n in ("REQ", "HITS", "MISSES", "MISSES_DUPLICATE"))
        wc = c.get("SQ_WAVE_CYCLES", 0) / b2
        print("   %-22s per book-step: icache req %8.0f hits %8.0f misses %7.0f (+dup %7.0f) = %.2f %% missed | ifetch %8.0f  fetch-level/wave-cycles %.3f  wait-LDS/wave-cycles %.3f  insts %6.0f"
              % (k, req, hit, miss, dup, 100.0 * (miss + dup) / max(req, 1), c.get("SQ_IFETCH", 0) / b2, c.get("SQ_IFETCH_LEVEL", 0) / b2 / max(wc, 1),
                 c.get("SQ_WAIT_INST_LDS", 0) / b2 / max(wc, 1), (c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_VALU", 0)) / b2))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
