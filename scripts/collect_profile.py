"""Copy the summaries of a scripts/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/<dst>/ and refresh the
per-kernel PMC entries of profiles/pmc_traffic.json (the ones bench.py replays into `roofline.traffic` / `roofline.issue`).
usage: collect_profile.py <tag> <dst>      e.g.  collect_profile.py r02b r02"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, dst = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out = os.path.join(ROOT, "profiles", dst)
os.makedirs(out, exist_ok=True)
for f in os.listdir(src):
    if f.startswith("bench_") and f.endswith(".json") or f == "pmc_summary.json":
        shutil.copy(os.path.join(src, f), os.path.join(out, f))
shutil.copy(os.path.join(src, "kt_C3", "kt_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_C3_steps200.csv"))
shutil.copy(os.path.join(src, "kt_C4", "kt_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_C4_shard_8192_steps200.csv"))
for kind in ("fetch", "write", "sq"):
    for B in (65536, 8192):
        shutil.copy(os.path.join(src, "pmc_%s_%d" % (kind, B), "p_counter_collection.csv"), os.path.join(out, "pmc_%s_%d.csv" % (kind, B)))
summ = json.load(open(os.path.join(src, "pmc_summary.json")))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
allp = json.load(open(path))
for B, key in ((65536, "C3"), (8192, "C3/8192")):
    for k, v in summ[str(B)].items():
        if k not in allp.get(key, {}) or "SQ_WAVES" not in v:
            continue
        e = allp[key][k]
        books = summ[str(B)]['k_step_batch']['SQ_WAVES']  # books per dispatch = event waves per dispatch (one per book)
        hbm = (v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024.0
        e.update(hbm_bytes_per_book_step=hbm / books, hbm_bytes_per_launch=hbm, fetch_size_kib_raw=v["FETCH_SIZE"],
                 write_size_kib=v["WRITE_SIZE"])
        e["source"] = "profiles/%s/pmc_fetch_%d.csv + pmc_write_%d.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; scripts/profile_round.sh)" % (dst, B, B)
        e["insts_per_book_step"] = dict(salu=v["SQ_INSTS_SALU"] / books, valu=v["SQ_INSTS_VALU"] / books, branch=v["SQ_INSTS_BRANCH"] / books,
                                        lds=v["SQ_INSTS_LDS"] / books,
                                        source="profiles/%s/pmc_sq_%d.csv (SQ_INSTS_* per dispatch / books per dispatch)" % (dst, B))
        print(key, k, {a: round(b, 1) for a, b in e["insts_per_book_step"].items() if a != "source"}, "hbm B/book-step", round(e["hbm_bytes_per_book_step"]))
json.dump(allp, open(path, "w"), indent=1)
