# Where k_step_events' HBM traffic goes (17.7 KB measured vs 13.1 KB algorithmic per book-step at 8 192
# books): FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload INGRESS` with the two suspects toggled -
#   the wave-parallel shuffle's lane-state record (BOURSE_AMD_EV_SEQ_SHUFFLE=1: draw by draw, no record) and
#   the order log (a -DBOURSE_AMD_EV_SKIP=2 build: no log writes; build_variants/libbourse_amd_nolog.so, built in the container).
# GPU box:  bash scripts/pmc_step_events.sh [books] [extra bench args]   -> gpurun_out/pmc_step_events/summary_<books>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-8192}; shift || true; XA="$*"
OUT=$R/gpurun_out/pmc_step_events; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
PA="--workload INGRESS --books $B --steps 24 --warmup 6 --no-cpu-baseline --preheat-steps 0 $XA"
for V in shipped noshuf nolog noshuf_nolog; do
  unset BOURSE_AMD_EV_SEQ_SHUFFLE BOURSE_AMD_LIBRARY
  case $V in noshuf*) export BOURSE_AMD_EV_SEQ_SHUFFLE=1;; esac
  case $V in *nolog) export BOURSE_AMD_LIBRARY=$R/build_variants/libbourse_amd_nolog.so;; esac
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C -d $OUT/${V}_$C -o p -f csv -- python3 $R/bench.py $PA > $OUT/${V}_$C.json 2> $OUT/${V}_$C.err
  done
done
unset BOURSE_AMD_EV_SEQ_SHUFFLE BOURSE_AMD_LIBRARY
python3 - <<PY | tee $OUT/summary_$B.txt
import collections, csv, glob, json
print("k_step_events / k_ingest HBM traffic per book-step at $B books (rocprofv3 --pmc FETCH_SIZE x 2 [gfx950] + WRITE_SIZE, KiB -> B), bench args: $PA")
for v in ("shipped", "noshuf", "nolog", "noshuf_nolog"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(float))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
                if k not in ("k_step_events", "k_ingest"): continue
                tot[k][c] += float(r["Counter_Value"]); n[k][c] += float(r["Grid_Size"]) / 64.0
    try:
        b = json.loads([l for l in open("$OUT/%s_FETCH_SIZE.json" % v) if l.startswith("{")][-1])
        alg = b["roofline"]["kernels"]; acct = b["roofline"].get("accounting")
    except Exception as e:
        alg, acct = {}, str(e)
    for k in sorted(tot):
        f = tot[k]["FETCH_SIZE"] / max(n[k]["FETCH_SIZE"], 1) * 2048.0; w = tot[k]["WRITE_SIZE"] / max(n[k]["WRITE_SIZE"], 1) * 1024.0
        a = alg.get(k, {}).get("bytes_per_book_step", float("nan"))
        print("%-13s %-14s fetch %7.0f B  write %7.0f B  total %7.0f B   algorithmic %7.0f B  ratio %.2f" % (v, k, f, w, f + w, a, (f + w) / a))
    if v == "shipped": print("   accounting of the run:", acct)
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
