"""Fold the summary of a scripts/pmc_all.sh run into the committed record:
  gpurun_out/pmc_<tag>/summary.json  ->  profiles/<round>/pmc_summary.json  (as measured, per config and kernel)
                                     ->  profiles/pmc_traffic.json          (what bench.py replays into roofline.traffic / .issue)
usage: pmc_merge.py <tag> <round> [--add]     e.g.  pmc_merge.py r04a r04;  --add: the round's record keeps its other configurations
(a later pmc_all.sh run of some configurations only)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_" + tag, "summary.json")))
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
if "--add" in sys.argv:
    summ = {**json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json"))), **summ}
json.dump(summ, open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json"), "w"), indent=1)
src = (f"profiles/{rnd}/pmc_summary.json (rocprofv3 --pmc passes of `bench.py --workload W --books B`, FETCH_SIZE / WRITE_SIZE / "
       f"SQ_INSTS_* / SQ_WAIT_* in separate passes, every counter divided by the book-steps of its dispatches: "
       f"scripts/pmc_all.sh, scripts/pmc_summarise.py)")
out = {"_format": "key = workload/books_per_gpu -> kernel -> per-book-step PMC figures; FETCH_SIZE x2 (gfx950 tallies 128-B "
                  "requests at 64 B: MI355X_MICROARCH.md, HBM), KiB -> bytes", "_source": src}
for key, cfg in summ.items():
    e = {}
    for k, v in cfg.items():
        if k.startswith("_"):
            continue
        p = v["per_book_step"]
        rec = {"launch": f"{k}, {v.get('book_steps_per_dispatch', 0):.0f} book-steps per dispatch (mean), "
                         f"{v['dispatches'].get('SQ_INSTS_SALU', 0)} dispatches"}
        if "hbm_bytes_per_book_step" in v:
            rec.update(hbm_bytes_per_book_step=v["hbm_bytes_per_book_step"], fetch_size_kib_raw_per_book_step=p["FETCH_SIZE"],
                       write_size_kib_per_book_step=p["WRITE_SIZE"])
        if "SQ_INSTS_SALU" in p:
            rec["insts_per_book_step"] = {"salu": p["SQ_INSTS_SALU"], "valu": p["SQ_INSTS_VALU"], "branch": p["SQ_INSTS_BRANCH"],
                                          "lds": p["SQ_INSTS_LDS"], "smem": p.get("SQ_INSTS_SMEM", 0.0), "vmem": p.get("SQ_INSTS_VMEM", 0.0)}
        if "wave_cycles" in v:
            wc = dict(v["wave_cycles"])
            if p.get("SQ_BUSY_CYCLES"):
                # SQ_WAVE_CYCLES sums the resident waves' QUAD-cycles (MI355X_MICROARCH.md, cycle constants), SQ_BUSY_CYCLES the busy
                # cycles of the 32 shader engines
                wc["resident_waves_avg"] = 4.0 * p["SQ_WAVE_CYCLES"] / (p["SQ_BUSY_CYCLES"] / 32.0)
                wc["waves_per_simd_avg"] = wc["resident_waves_avg"] / 1024.0
            if p.get("SQ_THREAD_CYCLES_VALU") and p.get("SQ_INSTS_VALU"):
                wc["valu_lane_utilisation"] = p["SQ_THREAD_CYCLES_VALU"] / (256.0 * p["SQ_INSTS_VALU"])
            wc["note"] = "the profiler serialises dispatches: each kernel ran ALONE on one part of the batch (no overlap with the other parts' kernels)"
            rec["occupancy"] = wc
        e[k] = rec
    if cfg.get("_bench", {}).get("accounting"):
        e["_accounting"] = cfg["_bench"]["accounting"]
    out[key] = e
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for key, e in out.items():
    if key.startswith("_"):
        continue
    for k, rec in e.items():
        if k.startswith("_"):
            continue
        i = rec.get("insts_per_book_step", {})
        print("%-10s %-22s hbm %7.0f B/book-step  scalar+branch %6.0f  vector %6.0f" % (
            key, k, rec.get("hbm_bytes_per_book_step", float("nan")), i.get("salu", 0) + i.get("branch", 0), i.get("valu", 0)))
