"""Inter-kernel gaps per stream from a rocprofv3 --kernel-trace CSV: how much of a part's per-step chain is launch gap.
usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by_q = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    by_q[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for q, ks in sorted(by_q.items()):
    ks.sort()
    ks = [k for k in ks if k[2] in ("k_agents_wave", "k_step_batch", "k_agents_fsm", "k_agents_mixed_wave")]
    if len(ks) < 50:
        continue
    ks = ks[len(ks) // 2:]  # the second half: steady state
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for a, b in zip(ks, ks[1:]):
        dur[a[2]].append(a[1] - a[0])
        gap[a[2] + "->" + b[2]].append(b[0] - a[1])
    span = (ks[-1][1] - ks[0][0]) / 1e3
    print("queue", q, "kernels", len(ks), "span %.0f us" % span)
    for k, v in dur.items():
        print("   %-22s avg %.1f us  (n=%d)" % (k, sum(v) / len(v) / 1e3, len(v)))
    for k, v in gap.items():
        v2 = sorted(v)
        print("   gap %-30s median %.1f us  avg %.1f us" % (k, v2[len(v2) // 2] / 1e3, sum(v) / len(v) / 1e3))
