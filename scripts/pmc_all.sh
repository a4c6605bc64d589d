# PMC record of EVERY bench configuration (round 4): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate
# passes), instruction counts and the issue / stall counters, per kernel, normalised per book-step.
# GPU box:  bash scripts/pmc_all.sh [tag] [configs...]      configs = workload:books, default: all of BASELINE's
#   -> gpurun_out/pmc_<tag>/summary.json   (scripts/pmc_merge.py <tag> <round> folds it into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
shift || true
CONFIGS=${*:-"C3:65536 C3:32768 C3:16384 C3:8192 C2:4096 C5:8192 C5M:8192"}
OUT=$R/gpurun_out/pmc_$TAG
FAILED=""
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the timed region is left alone (--profile-every 0, no repeats, no pre-heat: bk_warm's one 100-step launch would mix a
# second launch shape into the fused kernels' means); 20 steps per launch so that a fused launch is 20 book-steps per book
PA="--steps 40 --warmup 20 --steps-per-launch 20 --no-cpu-baseline --profile-every 0 --repeats 0 --preheat-steps 0"
PA_AGENTS=$PA
for C in $CONFIGS; do
  W=${C%%:*}; B=${C##*:}
  # (the external-agents stream fills the pools: at most 33 steps, bench.py bench_ingress)
  WL=$W
  if [ $W = INGRESS ]; then PA="--steps 24 --warmup 6 --no-cpu-baseline --preheat-steps 0"; else PA=$PA_AGENTS; fi
  # INGRESSMIX: the external-agents stream with modifications and market orders (round 6; bench.py keys its PMC record the same way)
  if [ $W = INGRESSMIX ]; then WL=INGRESS; PA="--steps 24 --warmup 6 --no-cpu-baseline --preheat-steps 0 --modify-frac 0.05 --market-frac 0.02"; fi
  run() { d=$1; shift; rocprofv3 --pmc "$@" -d $OUT/${W}_${B}_$d -o p -f csv -- python3 $R/bench.py --workload $WL --books $B $PA > $OUT/${W}_${B}_$d.json 2> $OUT/${W}_${B}_$d.err
          # a pass whose bench died (round 4: a NameError AFTER the timed region, stdout empty) must not pass silently
          grep -q '^{' $OUT/${W}_${B}_$d.json || { echo "pmc_all: $W:$B pass '$d' printed no bench line" >&2; grep -v "^W2\|rocprofiler" $OUT/${W}_${B}_$d.err | tail -n 8 >&2; FAILED="$FAILED ${W}_${B}_$d"; }; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run sq SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES
  run stall SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
done
python3 $R/scripts/pmc_summarise.py $OUT || FAILED="$FAILED summary"
# the raw per-dispatch CSVs are tens of thousands of lines each: only the summary travels back
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
if [ -n "$FAILED" ]; then echo "pmc_all: FAILED passes:$FAILED" >&2; exit 1; fi
echo "pmc_all: every pass printed its bench line"
