# Why is a kernel's longest launch 1.6 x its mean (k_agents_wave<8> at C5: 139 us mean, 224 us max)?
# rocprofv3 kernel trace of `bench.py --workload W`, then per kernel: the distribution of the launch durations and how they
# depend on WHAT ELSE RAN during the launch (every other kernel's overlap with it, from the trace's start / end stamps).
# GPU box:  bash scripts/launch_tail.sh [workload] [extra bench args]   -> gpurun_out/launch_tail/<workload>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-C5}; shift || true
OUT=$R/gpurun_out/launch_tail; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt_$W
rocprofv3 --kernel-trace -d $OUT/kt_$W -o kt -f csv -- python3 $R/bench.py --workload $W --steps 60 --warmup 20 --no-cpu-baseline --preheat-steps 0 --repeats 0 --profile-every 0 "$@" > $OUT/bench_$W.json 2> $OUT/kt_$W.err
python3 - <<PY | tee $OUT/$W.txt
import csv, glob, collections
import numpy as np
rows = []
for f in glob.glob("$OUT/kt_$W/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
        rows.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", 0) or 0)))
rows.sort(key=lambda x: x[1])
step_k = [r for r in rows if r[0] in ("k_agents_wave", "k_step_batch", "k_agents_fsm", "k_agents_mixed_wave", "k_run_wave")]
# steady part only: drop the first and last 10 % of the launches
n = len(step_k); step_k = step_k[n // 10: n - n // 10]
S = np.array([r[1] for r in step_k]); E = np.array([r[2] for r in step_k]); names = [r[0] for r in step_k]
print("workload $W: %d step-kernel launches in the steady part of the trace" % len(step_k))
for k in sorted(set(names)):
    idx = [i for i, nme in enumerate(names) if nme == k]
    d = (E[idx] - S[idx]) / 1e3
    # concurrency: for each launch, the time-weighted number of OTHER step kernels running during it, by kernel name
    conc = collections.defaultdict(list)
    for i in idx:
        ov = np.clip(np.minimum(E, E[i]) - np.maximum(S, S[i]), 0, None).astype(float)
        ov[i] = 0
        for k2 in sorted(set(names)):
            m = np.array([nme == k2 for nme in names])
            conc[k2].append(ov[m].sum() / max(E[i] - S[i], 1))
    print("%-22s n %4d  mean %6.1f us  sd %5.1f  min %6.1f  p50 %6.1f  p90 %6.1f  max %6.1f" % (k, len(d), d.mean(), d.std(), d.min(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
    order = np.argsort(d)
    q = max(1, len(d) // 5)
    for lab, sel in (("fastest fifth", order[:q]), ("slowest fifth", order[-q:])):
        print("    %-14s %6.1f us; other kernels running beside it (mean count): %s" % (lab, d[sel].mean(), {k2: round(float(np.mean(np.array(conc[k2])[sel])), 2) for k2 in conc}))
    cc = {k2: round(float(np.corrcoef(d, np.array(conc[k2]))[0, 1]), 2) for k2 in conc if np.std(conc[k2]) > 0}
    print("    correlation of the duration with the concurrency of:", cc)
PY
rm -rf $OUT/kt_$W
