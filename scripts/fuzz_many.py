import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
bad = 0
for seed in range(100, 220):
    try:
        T.test_fuzz_agent_sets_and_markets_vs_oracle(bk, oracle, seed)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAIL", str(e)[:300])
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300])
    except BaseException as e:  # pytest.skip: the drawn configuration overflowed the pool (flagged)
        if type(e).__name__ != "Skipped":
            raise
for seed in range(2000, 2060):
    try:
        T.test_fuzz_random_agent_configs_vs_oracle(bk, oracle, seed)
    except AssertionError as e:
        bad += 1; print("R seed", seed, "FAIL", str(e)[:300])
    except Exception as e:
        bad += 1; print("R seed", seed, "ERR", type(e).__name__, str(e)[:300])
    except BaseException as e:
        if type(e).__name__ != "Skipped":
            raise
print("done, failures:", bad)
