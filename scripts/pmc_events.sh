# instruction counts and wave-cycle split of k_step_events / k_ingest on the device-ingress workload (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-8192}; OUT=$R/gpurun_out/pmc_events; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES -d $OUT/sq -o p -f csv -- python3 $R/scripts/device_ingress_rate.py $B > $OUT/sq.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $OUT/st -o p -f csv -- python3 $R/scripts/device_ingress_rate.py $B > $OUT/st.txt 2>&1
python3 - <<PY | tee $OUT/summary_$B.txt
import collections, csv, glob
for kind in ("sq", "st"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % kind, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
            if k not in ("k_step_events", "k_ingest"): continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_INSTS_SALU", "SQ_WAVE_CYCLES"): n[k] += float(r["Grid_Size"]) / 64.0
    for k, c in agg.items():
        if kind == "sq":
            print(k, "per book-step:", {a[9:]: round(b / n[k]) for a, b in c.items() if a != "SQ_WAVES"})
        else:
            wc = c["SQ_WAVE_CYCLES"]
            print(k, "wave-cycles per book-step %.0f (x4 = clocks): issuing %.2f waiting-for-issue %.2f waiting(any) %.2f scalar %.2f valu %.2f wait-LDS %.2f" % (
                wc / n[k], c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_ACTIVE_INST_SCA"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_WAIT_INST_LDS"] / wc))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
