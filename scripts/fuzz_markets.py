"""Fuzz MarketEnv with RandomMarketAgents (ref crates/step_sim/src/market_env.rs:110-132, agents/random_agent.rs:122-220) over fresh
seeds against the oracle's ManyMarkets: 1-4 assets with random tick sizes, 1-6 groups on random assets (some assets nobody trades),
agent counts that land on every pool size (64 .. 512 slots), small and multi-part batches, launches in random chunks - L2 history of
every step and book, RNG states, trades and live orders of sample books (tests/test_gpu_parity.py _compare_markets).  Since round 5
the markets' step batches run on the keyed / assembly event loops (the other assets' events as events that do nothing).  GPU box."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T

lo, hi = int(os.environ.get("FUZZ_LO", 0)), int(os.environ.get("FUZZ_HI", 300))
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(880_000 + seed)
    A = int(rng.integers(1, 5))
    ticks = [int(rng.choice([1, 2, 3, 5])) for _ in range(A)]
    budget = int(rng.choice([40, 64, 100, 128, 200, 256, 400, 512]))
    n_groups = int(rng.integers(1, 7))
    sizes = rng.multinomial(budget - n_groups, np.ones(n_groups) / n_groups) + 1
    groups = []
    for n in sizes:
        a = int(rng.integers(0, A))
        t0 = int(rng.integers(5, 60)); v0 = int(rng.integers(1, 60))
        groups.append((a, int(n), (t0, t0 + int(rng.integers(1, 40))), (v0, v0 + int(rng.integers(1, 30))),
                       ticks[a] * int(rng.integers(1, 4)), float(rng.choice([0.05, 0.3, 0.6, 0.9, 1.0]))))
    big = rng.random() < 0.08
    NM = int(rng.integers(4200, 7000)) if big else int(rng.integers(2, 300))
    steps = int(rng.integers(2, 8)) if big else int(rng.integers(1, 40))
    cuts = sorted(set(int(x) for x in rng.integers(1, steps + 1, size=int(rng.integers(0, 3)))))
    chunks = [b - a for a, b in zip([0] + cuts, cuts + [steps]) if b > a]
    try:
        T._compare_markets(bk, oracle, NM, ticks, groups, int(rng.integers(1, 33)), steps, seed=int(rng.integers(1, 10_000)), chunks=chunks)
    except AssertionError as e:
        bad += 1; print("seed", seed, dict(NM=NM, ticks=ticks, groups=groups, steps=steps, chunks=chunks), "FAIL", str(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300], flush=True)
print(f"markets fuzz: {hi - lo} configurations, failures: {bad}")
