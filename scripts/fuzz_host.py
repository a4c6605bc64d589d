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

# extreme values: prices at both ends of u32, volumes that overflow the u32 side totals (the reference's `+=` wraps in
# release builds), zero volumes, market orders against empty sides, modifies to the same price / to extremes
def extreme_stream(seed, n_steps=25):
    rng = np.random.default_rng(900_000 + seed)
    tick = int(rng.choice([1, 2, 5]))
    top = (2**32 - 1) // tick * tick
    prices = [tick, 2 * tick, 3 * tick, top, top - tick, top - 2 * tick, (2**31 // tick) * tick, (2**31 // tick) * tick + tick]
    vols = [0, 1, 2, 7, 2**31, 2**32 - 1, 2**32 - 2, 123456789]
    g = bk.core.StepEnv(seed, 0, tick, 100_000)
    r = oracle.StepEnv(seed, 0, tick, 100_000)
    ids = []
    for _ in range(n_steps):
        for _k in range(int(rng.integers(0, 14))):
            u = rng.random()
            if u < 0.6 or not ids:
                side, vol = bool(rng.integers(0, 2)), int(rng.choice(vols))
                price = None if rng.random() < 0.15 else int(rng.choice(prices))
                a, b = g.place_order(side, vol, 3, price), r.place_order(side, vol, 3, price)
                assert a == b
                ids.append(a)
            elif u < 0.8:
                i = int(rng.choice(ids)); g.cancel_order(i); r.cancel_order(i)
            else:
                i = int(rng.choice(ids))
                np_ = None if rng.random() < 0.4 else int(rng.choice(prices))
                nv = None if rng.random() < 0.3 else int(rng.choice(vols))
                g.modify_order(i, np_, nv); r.modify_order(i, np_, nv)
        g.step(); r.step()
        assert np.array_equal(g.level_2_data_array(), r.level_2_data_array())
    assert g.get_trades() == r.get_trades()
    assert g.get_orders() == r.get_orders()

xbad = 0
for seed in range(lo, hi):
    try:
        extreme_stream(seed)
    except AssertionError as e:
        xbad += 1; print("extreme seed", seed, "FAIL", str(e)[:300])
    except Exception as e:
        xbad += 1; print("extreme seed", seed, "ERR", type(e).__name__, str(e)[:300])
print("extreme-value streams failed:", xbad, "of", hi - lo)
