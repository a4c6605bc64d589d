"""Fuzz campaign for k_agents_mixed_wave (wave_mixed.hpp): randomly drawn Noise / Momentum AgentSets on independent books,
launch chunkings cycling through all four member pipelines (incl. a checkpoint / restore into a fresh env), against the
oracle.  usage: fuzz_wave_members.py [first_seed [n_seeds [books_scale]]]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BOURSE_FUZZ_WAVE_MEMBERS"] = "1"
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
if len(sys.argv) > 3:
    os.environ["BOURSE_FUZZ_BOOKS_SCALE"] = sys.argv[3]
import bourse_amd as bk, pyoracle as oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402

bad = skipped = 0
for seed in range(first, first + n):
    try:
        T.test_fuzz_agent_sets_and_markets_vs_oracle(bk, oracle, seed, checkpoint_at=seed)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAIL", str(e)[:400], flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300], flush=True)
    except BaseException as e:  # pytest.skip: the drawn configuration overflowed the pool (flagged)
        if type(e).__name__ != "Skipped":
            raise
        skipped += 1
print(f"done: seeds {first}..{first + n - 1}, failures {bad}, skipped {skipped}")
