"""Re-run one fuzz seed with full diagnostics (GPU box): fuzz_one.py <members|random> <seed> [pipeline-override]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
kind, seed = sys.argv[1], int(sys.argv[2])
fn = T.test_fuzz_agent_sets_and_markets_vs_oracle if kind == "members" else T.test_fuzz_random_agent_configs_vs_oracle
try:
    fn(bk, oracle, seed)
    print("seed", seed, "ok")
except Exception as e:
    import traceback
    traceback.print_exc()
    print(str(e)[:3000])
