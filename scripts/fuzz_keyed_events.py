"""Fuzz the host-driven step on the keyed loop (step_events.hpp step_events_keyed) over fresh seeds: every book of small
batches against its own oracle env - level 2 of every step, every trade, the whole order log - on random mixes of clean
steps (new / cancel / modify / market orders: the keyed form) and steps that must fall back (volume 0, more events
than pool slots, prices outside the key window, full pools), all four pool sizes, three tick sizes, narrow and wide price
ranges, ordinary and extreme volumes.  Prints how many book-steps ran keyed.  GPU box.  FUZZ_LO / FUZZ_HI."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bourse_amd as bk, pyoracle as oracle
import test_gpu_keyed_events as K

lo, hi = int(os.environ.get("FUZZ_LO", 0)), int(os.environ.get("FUZZ_HI", 300))
bad = busy_n = clean_n = keyed_n = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(660_000 + seed)
    pool = int(rng.choice([64, 128, 256, 512]))
    n_max = int(rng.integers(1, {64: 16, 128: 40, 256: 70, 512: 110}[pool]))
    if rng.random() < 0.05:
        n_max = pool + int(rng.integers(1, 20))  # some steps hold more events than the pool has slots
    B, T = int(rng.integers(1, 40)), int(rng.integers(1, 15))
    tick = int(rng.choice([1, 2, 5]))
    centre = int(rng.choice([100, 100, 20_000, 3_000_000, (2**32 - 1) // tick - 40]))
    width = int(rng.choice([3, 8, 8, 30, 40_000]))  # 40 000 ticks: wider than the key window
    lo_p, hi_p = max(1, centre - width), min((2**32 - 1) // tick, centre + width + 1)
    kw = dict(p_market=float(rng.choice([0.0, 0.01, 0.05])), p_mod=float(rng.choice([0.0, 0.003, 0.03, 0.1, 0.25])),
              p_zero=float(rng.choice([0.0, 0.0, 0.002, 0.02])), tick=tick, lo=lo_p, hi=hi_p,
              vols=[1, 2, 7, 2**31, 2**32 - 1, 2**32 - 2, 123456789] if rng.random() < 0.15 else None,
              # a fifth of the narrow-range configurations well above price 40 000 also get stink bids far below the book: the
              # top-anchored key window with saturated bids (round 6)
              far=((0.06, 1, 60) if rng.random() < 0.5 else (0.06, 20 * centre // 19, 20 * centre // 19 + 60))
              if (rng.random() < 0.25 and width <= 30 and centre == 3_000_000) else None)
    try:
        env, refs, busy, clean = K._drive(bk, oracle, pool, n_max, B, T, 1000 + seed, **kw)
        try:
            K._same_as_oracle(env, refs, allow_flags=n_max > pool // 3)
            keyed = env.event_steps_keyed()
            assert np.all(keyed <= clean.sum(axis=0)), "a step outside the keyed form ran keyed"
            busy_n += int(busy.sum()); clean_n += int(clean.sum()); keyed_n += int(keyed.sum())
        finally:
            env.close()
    except AssertionError as e:
        bad += 1; print("seed", seed, dict(pool=pool, n_max=n_max, B=B, T=T, **kw), "FAIL", str(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300], flush=True)
print(f"keyed-events fuzz: {hi - lo} configurations, failures: {bad}; {busy_n} book-steps with events, {clean_n} of the keyed form by their calls, "
      f"{keyed_n} ran keyed")
