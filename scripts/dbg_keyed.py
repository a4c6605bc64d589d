import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bourse_amd as bk, pyoracle as oracle
import test_gpu_parity as T
groups = T.C2_GROUPS; n_books = 96; n_steps = 40; levels = 16
n_agents = sum(g[0] for g in groups)
env = bk.ManyBookEnv(n_books, 101, 0, 2, 100_000, True, levels=levels, max_live_orders=n_agents, trade_capacity=2*n_agents*n_steps, history_capacity=n_steps)
env.set_random_agents(groups); env.set_pipeline("split"); env.strict = False
env.run(n_steps)
ref = oracle.ManyBooks(n_books, 101, 0, 2, 100_000, True, levels, groups); ref.run(n_steps, n_threads=4)
h, w = env.history(), ref.history()
bad = np.argwhere(h != w)
print("flags", np.unique(env.flags()), "mismatches", len(bad))
if len(bad):
    s, b, _ = bad[0]
    print("first at step", s, "book", b)
    print("dev", h[s, b][:12]); print("ref", w[s, b][:12])
    got = env.trades(b, first=0); exp = ref.book(b).trades_array()
    t0 = s * 100_000
    g = got[(got["t"] >= t0) & (got["t"] < t0 + 100_000)]; e = exp[(exp["t"] >= t0) & (exp["t"] < t0 + 100_000)]
    print("dev trades", len(g)); print(g)
    print("ref trades", len(e)); print(e)
    # the previous step's resting orders (oracle): who was at the touch
    books_bad = sorted(set(int(x[1]) for x in bad if x[0] == s)); print("books bad at that step", books_bad[:20])
