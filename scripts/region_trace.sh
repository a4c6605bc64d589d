# Kernel trace of a bench.py command line: k_agents_fsm / k_step_batch durations over time (buckets of 30 launches each) and the
# idle gaps between launches - where does the first timed region lose its 5-8 %?  GPU box.
# usage: region_trace.sh <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/region_trace
rocprofv3 --kernel-trace -d $R/gpurun_out/region_trace -o kt -f csv -- python3 $R/bench.py --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/region_trace.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/region_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t00 = int(rows[0]["Start_Timestamp"])
step = [r for r in rows if "k_agents_fsm" in r["Kernel_Name"] or "k_step_batch" in r["Kernel_Name"]]
last_end = None; bucket = []
def flush():
    if not bucket: return
    fsm = [e - s for n, s, e in bucket if "fsm" in n]; stp = [e - s for n, s, e in bucket if "step_batch" in n]
    print(f"t={ (bucket[0][1]-t00)/1e6:9.2f} ms  {len(bucket):3d} launches  span {(max(e for _,_,e in bucket)-bucket[0][1])/1e3:8.1f} us  fsm {sum(fsm)/max(len(fsm),1)/1e3:6.1f}  step {sum(stp)/max(len(stp),1)/1e3:6.1f}")
    bucket.clear()
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 200_000:
        flush(); print(f"   -- gap {(s-last_end)/1e3:.0f} us; other kernels in it: {sorted(set(x['Kernel_Name'][:40] for x in rows if last_end <= int(x['Start_Timestamp']) < s))[:6]}")
    bucket.append((r["Kernel_Name"], s, e)); last_end = max(last_end or 0, e)
    if len(bucket) == 30: flush()
flush()
PY
