"""Where a step of the host-array ingress spends its HOST time (scripts/host_driven_rate.py's `tickets` loop, call by call).
GPU box:  python scripts/host_ingress_profile.py [books]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bourse_amd as bk
B, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 48, 40
n = B * N
off = (np.arange(B + 1, dtype=np.uint64) * N)
rng = np.random.default_rng(0)
batch = (np.ones(n, np.uint32), rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(1, 30, size=n).astype(np.uint32),
         np.zeros(n, np.uint32), rng.integers(90, 111, size=n).astype(np.uint32), np.zeros(n, np.uint64))
env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=512, max_orders=0, trade_capacity=64, strict=False, history_capacity=0)
env.enable_device_ingress(N)
IDS, ST = np.empty(n, np.uint64), np.empty((B, 2), np.uint32)
acc = {"submit": 0.0, "step": 0.0, "result": 0.0, "clear": 0.0}
prev = None
for s in range(T + 5):
    if s == 5:
        env.sync(); acc = dict.fromkeys(acc, 0.0); t00 = time.perf_counter()
    t0 = time.perf_counter()
    t = env.submit_instructions_all_async(off, batch)
    t1 = time.perf_counter()
    env.step(sync=False)
    t2 = time.perf_counter()
    if prev is not None:
        env.submit_result(prev, out=IDS, status=ST)
    t3 = time.perf_counter()
    env.clear_trades()
    t4 = time.perf_counter()
    prev = t
    acc["submit"] += t1 - t0; acc["step"] += t2 - t1; acc["result"] += t3 - t2; acc["clear"] += t4 - t3
env.sync()
tot = time.perf_counter() - t00
print(f"B={B}: {tot / T * 1e3:.3f} ms/step = {B * T / tot / 1e6:.1f} M book-steps/s; host time per call: " +
      ", ".join(f"{k} {v / T * 1e3:.3f} ms" for k, v in acc.items()))
# the host copy alone: the same 25 B per element into ordinary memory with numpy (one thread)
dst = [np.empty_like(a) for a in batch]
t0 = time.perf_counter()
for _ in range(10):
    for d, a in zip(dst, batch):
        np.copyto(d, a)
print(f"numpy copy of one batch ({sum(a.nbytes for a in batch) / 1e6:.1f} MB), one thread: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
for nt in (1, 4, 8, 16):
    os.environ["BOURSE_AMD_HOST_THREADS"] = str(nt)
    e2 = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=512, max_orders=0, trade_capacity=64, strict=False, history_capacity=0)
    e2.enable_device_ingress(N)
    for s in range(3):
        e2.submit_instructions_all_async(off, batch); e2.step(sync=False)
    e2.sync()
    t0 = time.perf_counter()
    for s in range(20):
        e2.submit_instructions_all_async(off, batch)
    dt = (time.perf_counter() - t0) / 20
    e2.sync()
    print(f"  BOURSE_AMD_HOST_THREADS={nt}: submit call {dt * 1e3:.3f} ms (no step in between: includes waiting for the slot)")
    e2.close()
