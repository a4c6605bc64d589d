"""Per-kernel, per-book-step summary of the rocprofv3 --pmc passes of scripts/pmc_all.sh (run on the GPU box).
usage: pmc_summarise.py <dir with WORKLOAD_BOOKS_{fetch,write,sq,stall}/ trees>  ->  <dir>/summary.json

Every counter is normalised by the BOOK-STEPS of the dispatches it was read on (sum over dispatches / sum of their
book-steps), taken from each dispatch's grid size - so launches of different shapes (parts of unequal size, the bench's
stand-alone full-batch launches) do not skew a mean:
  lane-per-book kernels (k_agents_fsm, k_agents_mixed_lanes): books = grid size
  wave-per-book kernels: books = grid size / 64;  the fused ones (k_run_*) step each book `spl` times per launch."""
import collections
import csv
import glob
import json
import os
import re
import sys

out_dir = sys.argv[1]
SPL = 20  # --steps-per-launch of scripts/pmc_all.sh
LANE_KERNELS = ("k_agents_fsm", "k_agents_mixed_lanes")
FUSED = ("k_run_random", "k_run_wave", "k_run_mixed")


def short(name):
    return name.split("(")[0].split("<")[0].split("::")[-1].strip()


def book_steps(kernel, grid):
    if kernel in LANE_KERNELS:
        return grid
    return grid / 64.0 * (SPL if kernel in FUSED else 1)


summary = {}
missing_bench = []
for d in sorted(glob.glob(os.path.join(out_dir, "*_*_sq"))):
    m = re.match(r"(.+)_(\d+)_sq$", os.path.basename(d))
    wl, books = m.group(1), int(m.group(2))
    key = f"{wl}/{books}"
    cfg = {}
    for kind in ("fetch", "write", "sq", "stall"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        bs = collections.defaultdict(lambda: collections.defaultdict(float))
        nd = collections.defaultdict(lambda: collections.defaultdict(int))
        for f in glob.glob(os.path.join(out_dir, f"{wl}_{books}_{kind}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith("k_") or k in ("k_delay", "k_gather_header", "k_book_service", "k_stats", "k_flags_summary"):
                    continue
                c = r["Counter_Name"]
                agg[k][c] += float(r["Counter_Value"])
                bs[k][c] += book_steps(k, float(r["Grid_Size"]))
                nd[k][c] += 1
        for k in agg:
            e = cfg.setdefault(k, {"per_book_step": {}, "dispatches": {}})
            for c in agg[k]:
                e["per_book_step"][c] = agg[k][c] / bs[k][c]
                e["dispatches"][c] = nd[k][c]
                e.setdefault("book_steps_per_dispatch", bs[k][c] / nd[k][c])
    try:
        line = [l for l in open(os.path.join(out_dir, f"{wl}_{books}_sq.json")) if l.startswith("{")][-1]
        b = json.loads(line)
        cfg["_bench"] = {"value": b["value"], "pipeline": b["config"]["pipeline"], "ms_per_step": b["ms_per_step"],
                         # the run's own event / trade / order rates and shape: what bench.algorithmic_bytes() is evaluated on
                         # (tests/test_roofline_accounting.py: algorithmic <= 1.05 x these passes' measured traffic)
                         "accounting": b["roofline"].get("accounting"),
                         "note": "the bench line of the SQ_INSTS pass (under the profiler: slower than an unprofiled run)"}
    except Exception as ex:  # noqa: BLE001
        cfg["_bench"] = {"error": str(ex)}
        missing_bench.append(key)
    # derived: HBM bytes (FETCH_SIZE counts 64-B units as 32 B on gfx950 -> x2; both in KiB), the wave-cycle split
    for k, e in cfg.items():
        if k.startswith("_"):
            continue
        p = e["per_book_step"]
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            e["hbm_bytes_per_book_step"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        if "SQ_WAVE_CYCLES" in p and p["SQ_WAVE_CYCLES"] > 0:
            wc = p["SQ_WAVE_CYCLES"]
            e["wave_cycles"] = {
                "issuing_frac": p.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                "waiting_for_issue_frac": p.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                "in_waitcnt_frac": p.get("SQ_WAIT_ANY", 0.0) / wc,
                "scalar_issuing_frac": p.get("SQ_ACTIVE_INST_SCA", 0.0) / wc,
                "valu_issuing_frac": p.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
            }
    summary[key] = cfg
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
for key, cfg in summary.items():
    print(key, cfg.get("_bench"))
    for k, e in cfg.items():
        if k.startswith("_"):
            continue
        p = e["per_book_step"]
        print("   %-22s hbm %7.0f B  salu %6.0f  branch %6.0f  valu %6.0f  lds %5.0f   %s" % (
            k, e.get("hbm_bytes_per_book_step", float("nan")), p.get("SQ_INSTS_SALU", float("nan")),
            p.get("SQ_INSTS_BRANCH", float("nan")), p.get("SQ_INSTS_VALU", float("nan")), p.get("SQ_INSTS_LDS", float("nan")),
            {a: round(b, 3) for a, b in e.get("wave_cycles", {}).items()}))
if missing_bench:  # round 4 shipped a summary whose every config had lost its bench line to a crash after the timed region
    sys.exit("pmc_summarise: no bench line for %s - the profiled command died; see the passes' .err files" % ", ".join(missing_bench))
