"""The headline configuration at full size for T steps (default 200 = the bench's timed region), every book's level-2
history, trade count and RNG state against the oracle on all host threads (GPU box)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bourse_amd as bk, pyoracle as oracle
B, T = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
CH = 25
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=32, max_live_orders=128, trade_capacity=64 * CH, history_capacity=CH)
env.set_random_agents(G)
ref = oracle.ManyBooks(B, 101, 0, 2, 100_000, True, 32, G)
tg = to = 0.0
for s in range(0, T, CH):
    env.clear_history(); env.clear_trades()
    t0 = time.perf_counter(); env.run(CH); env.sync(); tg += time.perf_counter() - t0
    t0 = time.perf_counter(); ref.run(CH, os.cpu_count() or 8); to += time.perf_counter() - t0
    h, w = env.history(), ref.history()[s:s + CH]
    assert np.array_equal(h, w), f"history differs in steps [{s}, {s + CH})"
assert not env.flags().any()
assert np.array_equal(env.trade_counts(), ref.trade_counts())
want = ref.rng_states()
got = np.array([env.rng_state(b) for b in range(0, B, 257)], dtype=np.uint64)
assert np.array_equal(got, want[::257])
print(f"C3 full size: {B} books x {T} steps bit-exact vs the oracle ({int(ref.trade_counts().sum())} trades); gpu {tg:.2f} s, oracle {to:.1f} s")
