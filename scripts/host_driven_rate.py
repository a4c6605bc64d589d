"""Host-driven path at scale: every step an external agent layer submits N instructions per book for ALL books with one
bk_submit_instructions_csr call, then bk_step (upload 16 B/event, shuffle + match on the device).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bourse_amd as bk
B, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 48, 30
env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=256, max_orders=N * (T + 5), trade_capacity=64 * (T + 5), strict=False,
                     history_capacity=0)
rng = np.random.default_rng(0)
off = (np.arange(B + 1, dtype=np.uint64) * N)
n = B * N
t_sub = t_step = 0.0
for s in range(T + 3):
    action = np.ones(n, dtype=np.uint32)
    ids = np.zeros(n, dtype=np.uint64)
    if s:
        canc = rng.random(n) < 0.3
        action[canc] = 2
        ids[canc] = rng.integers(0, s * N * 0.6, size=int(canc.sum()))  # ids created in earlier steps (>= 0.7 N placed per step)
    sides = rng.integers(0, 2, size=n).astype(bool)
    vols = rng.integers(1, 30, size=n).astype(np.uint32)
    traders = np.zeros(n, dtype=np.uint32)
    prices = rng.integers(90, 111, size=n).astype(np.uint32)
    t0 = time.perf_counter()
    env.submit_instructions_all(off, (action, sides, vols, traders, prices, ids))
    t1 = time.perf_counter()
    env.step()
    t2 = time.perf_counter()
    if s >= 3:
        t_sub += t1 - t0; t_step += t2 - t1
print(f"B={B} x {N} instructions/book/step: submit {t_sub / T * 1e3:.1f} ms/step ({n / (t_sub / T) / 1e6:.1f} M instr/s), "
      f"bk_step {t_step / T * 1e3:.1f} ms/step; {B * T / (t_sub + t_step) / 1e6:.2f} M book-steps/s, flags {np.unique(env.flags())}, "
      f"trades/book-step {env.trade_counts().sum() / (B * (T + 3)):.1f}")
env.profile(1)
for s in range(5):
    env.submit_instructions_all(off, (np.ones(n, dtype=np.uint32), sides, vols, traders, prices, ids))
    env.step()
ms, nl = env.profile_read_kind(3)
print(f"k_step_events: {ms / max(nl, 1):.3f} ms per launch ({nl} launches), pool registers R = {env._L.bk_state_bytes_per_book(env._h) // 4 // 320}")
