"""Fuzz the RandomAgents parity test over fresh seeds (pipelines fused / split / mixed / wave_split / wave and the wave
decode's look-ahead rotate with the seed).  GPU box.  FUZZ_LO / FUZZ_HI select the seed range."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
bad = n = 0
for seed in range(int(os.environ.get("FUZZ_LO", 10000)), int(os.environ.get("FUZZ_HI", 12000))):
    n += 1
    try:
        T.test_fuzz_random_agent_configs_vs_oracle(bk, oracle, seed)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAIL", str(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300], flush=True)
print("done:", n, "configurations, failures:", bad)
