"""Fuzz both randomised parity tests over fresh seeds and several part configurations of the split pipeline (GPU box).
FUZZ_LO / FUZZ_HI select the seed range."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
bad = skipped = 0
for parts, mp in (("2", "64"), ("5", "64"), ("3", "4096")):
    os.environ["BOURSE_AMD_SPLIT_PARTS"] = parts; os.environ["BOURSE_AMD_MIN_PART"] = mp
    for seed in range(int(os.environ.get('FUZZ_LO', 5000)), int(os.environ.get('FUZZ_HI', 5150))):
        for fn in (T.test_fuzz_agent_sets_and_markets_vs_oracle, T.test_fuzz_random_agent_configs_vs_oracle):
            try:
                fn(bk, oracle, seed)
            except AssertionError as e:
                bad += 1; print(parts, mp, fn.__name__, "seed", seed, "FAIL", str(e)[:300])
            except Exception as e:
                bad += 1; print(parts, mp, fn.__name__, "seed", seed, "ERR", type(e).__name__, str(e)[:300])
            except BaseException as e:  # pytest.skip: the drawn configuration overflowed the 512-slot pool (flagged)
                if type(e).__name__ != "Skipped":
                    raise
                skipped += 1
print("done, failures:", bad, "skipped (pool overflow flagged):", skipped)
