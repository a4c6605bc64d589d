"""Debug helper (GPU box): dump first differences between the HIP path and the oracle for a tiny case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bourse_amd, pyoracle
groups = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]
B, T, L = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 16
env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, True, levels=L, max_live_orders=64, trade_capacity=4096, history_capacity=T)
env.set_random_agents(groups); env.run(T)
ref = pyoracle.ManyBooks(B, 101, 0, 2, 100_000, True, L, groups); ref.run(T, 1)
print("flags", env.flags())
h, w = env.history(), ref.history()
for s in range(T):
    for b in range(B):
        if not np.array_equal(h[s, b], w[s, b]):
            print("step", s, "book", b, "\n gpu", h[s, b][:13], "\n ref", w[s, b][:13]); break
print("rng gpu", [hex(x) for x in env.rng_state(0)], "ref", [hex(int(x)) for x in ref.rng_states()[0]])
print("trade counts", env.trade_counts(), ref.trade_counts(), "orders ref", ref.order_counts())
gt, rt = env.trades(0, first=0), ref.book(0).trades_array()
for i in range(min(len(gt), len(rt), 12)):
    print(i, "gpu", tuple(int(gt[f][i]) for f in gt.dtype.names), "ref", tuple(int(rt[f][i]) for f in rt.dtype.names))
o = ref.book(0).orders_array()
print("ref orders (first 12):")
for r in o[:12]: print("  ", tuple(int(r[f]) for f in o.dtype.names))
lo = env.live_orders(0)
print("gpu live:", len(lo), "ref active:", int((o["status"] == 1).sum()))
for r in lo[:12]: print("  ", int(r["order_id"]), int(r["side"]), int(r["price"]), int(r["vol"]))
