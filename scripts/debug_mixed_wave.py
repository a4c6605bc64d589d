"""First-light check of k_agents_mixed_wave (wave_mixed.hpp) against the oracle, with a readable diff."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bourse_amd as bk  # noqa: E402
import pyoracle  # noqa: E402

MOM = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=5.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0,
           price_dist_sigma=10.0)
NOISE = dict(tick_size=2, p_limit=0.3, p_market=0.2, p_cancel=0.2, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
CASES = {
    "noise20": (8, [("noise", 0, 20, NOISE)], 10, 30, 1, 256),
    "doc": (16, [("momentum", 0, 10, dict(MOM, tick_size=1)), ("noise", 10, 20, dict(NOISE, tick_size=1))], 10, 50, 1, 256),
    "c5m": (6, [("momentum", 0, 256, dict(MOM, demand=20.0)), ("noise", 256, 256, NOISE)], 64, 30, 2, 512),
}
for name in sys.argv[1:] or list(CASES):
    B, members, levels, T, tick, pool = CASES[name]
    env = bk.ManyBookEnv(B, 101, 0, tick, 1_000_000, True, levels=levels, max_live_orders=pool, trade_capacity=64 * T * 8,
                         history_capacity=T, strict=False)
    env.set_agents(members)
    env.set_pipeline("wave_split")
    print(name, env.pipeline())
    env.run(T)
    ref = pyoracle.ManyBooks(B, 101, 0, tick, 1_000_000, True, levels, members=members)
    ref.run(T, 2)
    h, w = env.history(), ref.history()
    ok = np.array_equal(h, w)
    print(" flags", np.unique(env.flags()), "history equal:", ok, "trades", int(env.trade_counts().sum()), int(ref.trade_counts().sum()))
    if not ok:
        bad = np.argwhere(h != w)
        print("  first diff (step, book, word):", bad[0], h[tuple(bad[0])], w[tuple(bad[0])], "n diffs", len(bad))
        s0 = bad[0][0]
        print("  gpu ", h[s0, bad[0][1], :13])
        print("  ref ", w[s0, bad[0][1], :13])
    wr = ref.rng_states()
    bad_rng = [b for b in range(B) if env.rng_state(b) != (int(wr[b, 0]), int(wr[b, 1]))]
    print(" rng mismatches:", bad_rng[:8])
    env.close()
