"""Raw C-ABI cost of bk_submit_instructions_csr (host half of Env for a whole batch) vs. BOURSE_AMD_HOST_THREADS.  GPU box."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bourse_amd as bk
from bourse_amd import _lib
B, N, T = 8192, 48, 20
env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=256, max_orders=N * (T + 5), trade_capacity=64 * (T + 5), history_capacity=0, strict=False)
rng = np.random.default_rng(0)
off = (np.arange(B + 1, dtype=np.uint64) * N); n = B * N
tw = tc = 0.0
for s in range(T + 3):
    action = np.ones(n, dtype=np.uint32); ids = np.zeros(n, dtype=np.uint64)
    sides = rng.integers(0, 2, size=n).astype(np.uint8); vols = rng.integers(1, 30, size=n).astype(np.uint32)
    traders = np.zeros(n, dtype=np.uint32); prices = rng.integers(90, 111, size=n).astype(np.uint32)
    out = np.full(n, 2**64 - 1, dtype=np.uint64); done = C.c_size_t(0)
    t0 = time.perf_counter()
    rc = env._L.bk_submit_instructions_csr(env._h, _lib.p64(off), _lib.p32(action), _lib.p8(sides), _lib.p32(vols), _lib.p32(traders), _lib.p32(prices), _lib.p64(ids), _lib.p64(out), C.byref(done))
    t1 = time.perf_counter()
    assert rc == 0
    env.step()
    if s >= 3: tc += t1 - t0
print(f"raw C call: {tc / T * 1e3:.2f} ms per step ({n / (tc / T) / 1e6:.0f} M instr/s), threads={os.environ.get('BOURSE_AMD_HOST_THREADS','default')}")
