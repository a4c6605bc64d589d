"""Fold a scripts/profile_round4.sh run into the committed record: bench lines, kernel stats and rate logs to
profiles/<round>/, the PMC summary through scripts/pmc_merge.py.   usage: collect_round4.py <tag> <round>"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
for f in sorted(os.listdir(src)):
    if (f.startswith("bench_") and f.endswith(".json")) or f.startswith("kernel_stats_") or f.endswith("_rate.txt") or f.endswith("_profile.txt") or f == "mixed_issue_bench.txt":
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
        print("copied", f)
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_merge.py"), tag, rnd], check=True)
