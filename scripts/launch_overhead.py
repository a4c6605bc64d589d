"""How long the HOST takes to enqueue a bk_run launch vs how long the GPU works on it (is the per-step launch loop launch-bound?).
GPU box:  python scripts/launch_overhead.py [books]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd as bk
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
G = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=32, max_live_orders=128, trade_capacity=64 * 20, history_capacity=20, strict=False)
env.set_random_agents(G)
env.run(40); env.clear_history(); env.clear_trades()
for n in (20, 20, 20, 100):
    env.clear_history(); env.clear_trades(); env.sync()
    t0 = time.perf_counter(); env.run(n, sync=False); t1 = time.perf_counter(); env.sync(); t2 = time.perf_counter()
    pipe, parts = env.pipeline()
    print(f"{B} books, {pipe} x {parts}: bk_run({n}) returned after {1e3 * (t1 - t0):.3f} ms (host enqueue of ~{2 * parts * n} kernels = "
          f"{1e6 * (t1 - t0) / (2 * parts * n):.1f} us each), the GPU finished after {1e3 * (t2 - t0):.3f} ms")
