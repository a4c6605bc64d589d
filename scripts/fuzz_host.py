"""Fuzz the host-driven path over fresh seeds: single-book place/cancel/modify streams (StepEnv) and multi-asset market
streams, against the oracle (GPU box).  FUZZ_LO / FUZZ_HI select the seed range."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
import pathlib, tempfile
bad = 0
lo, hi = int(os.environ.get("FUZZ_LO", 100)), int(os.environ.get("FUZZ_HI", 400))
for seed in range(lo, hi):
    for fn in (T.test_host_driven_random_stream_matches_oracle, T.test_market_host_driven_random_stream):
        try:
            fn(bk, oracle, seed)
        except AssertionError as e:
            bad += 1; print(fn.__name__, "seed", seed, "FAIL", str(e)[:300])
        except Exception as e:
            bad += 1; print(fn.__name__, "seed", seed, "ERR", type(e).__name__, str(e)[:300])
    try:  # immediate-mode OrderBook: JSON snapshot vs the oracle's, cross-loaded, continued on all four books
        with tempfile.TemporaryDirectory() as d:
            T.test_json_snapshot_matches_oracle_and_round_trips(bk, oracle, pathlib.Path(d), seed)
    except AssertionError as e:
        bad += 1; print("json seed", seed, "FAIL", str(e)[:300])
    except Exception as e:
        bad += 1; print("json seed", seed, "ERR", type(e).__name__, str(e)[:300])
print("done, failures:", bad, "of", 3 * (hi - lo))

# streaming egress (L2 ring + compacted trade stream) over random shapes: T not a multiple of the chunk, tiny and
# multi-part batches, every pipeline
import numpy as np
srng = np.random.default_rng(lo)
sbad = 0
for i in range(max(20, (hi - lo) // 10)):
    shape = (int(srng.integers(1, 200)), int(srng.integers(1, 40)), int(srng.integers(1, 9)), str(srng.choice(["auto", "fused", "split"])))
    try:
        T.test_streaming_l2_and_trades_together(bk, oracle, shape)
    except AssertionError as e:
        sbad += 1; print("stream", shape, "FAIL", str(e)[:300])
    except Exception as e:
        sbad += 1; print("stream", shape, "ERR", type(e).__name__, str(e)[:300])
print("streaming shapes failed:", sbad)
