"""External agent layers at scale: every step an agent layer hands over N instructions per book for ALL books as six HOST
numpy arrays (the reference's BaseNumpyAgent contract: src/bourse/step_sim/agents/base_agent.py:67-116, runner.py:103-112,
rust/src/step_sim_numpy.rs:233-275), then the books step.  Four ways through the library, same workload, same box:

  host-env   bk_submit_instructions_csr + bk_step: the host half of Env walks the arrays (tick check, ids, queues on CPU
             threads), 16 B per event uploaded (rounds 1-4's only way for host arrays)
  sync       bk_submit_instructions_host on a device-ingress env, ids fetched before the step (the reference's call shape)
  tickets    the same with the ids of step s fetched after step s + 1 went out (two submits in flight)
  views      tickets with the results read in place (read-only views of the pinned staging, no copy out)
  staging    tickets + the arrays written in place into the library's pinned staging (no host copy inside the library)

The instruction arrays are generated ahead of the timed loops: the rates are the library's, not numpy's RNG.
GPU box:  python scripts/host_driven_rate.py [books]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bourse_amd as bk
B, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 48, 30
POOL = 512 if B > 8192 else 256
n = B * N
off = (np.arange(B + 1, dtype=np.uint64) * N)
rng = np.random.default_rng(0)


def make(s):
    action = np.ones(n, dtype=np.uint32)
    ids = np.zeros(n, dtype=np.uint64)
    if s:
        canc = rng.random(n) < 0.3
        action[canc] = 2
        ids[canc] = rng.integers(0, s * N * 0.6, size=int(canc.sum()))  # ids created in earlier steps (>= 0.7 N placed per step)
    return (action, rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(1, 30, size=n).astype(np.uint32),
            np.zeros(n, dtype=np.uint32), rng.integers(90, 111, size=n).astype(np.uint32), ids)


batches = [make(s) for s in range(T + 3)]
IDS, ST = np.empty(n, dtype=np.uint64), np.empty((B, 2), dtype=np.uint32)  # the consumer's own result arrays, reused
KEYS = ("action", "side", "vol", "trader_id", "price", "order_id")


def env_new(ingress):
    e = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=POOL, max_orders=N * (T + 5), trade_capacity=64 * (T + 5),
                       strict=False, history_capacity=0)
    if ingress:
        e.enable_device_ingress(N)
    return e


def report(name, env, dt, extra=""):
    f = np.unique(env.flags())
    print(f"{name:9s} B={B} x {N} instructions/book/step: {dt / T * 1e3:.3f} ms/step -> {B * T / dt / 1e6:.2f} M book-steps/s "
          f"({n * T / dt / 1e6:.0f} M instructions/s), flags {f}, trades/book-step {env.trade_counts().sum() / (B * (T + 3)):.1f}, keyed steps {env.event_steps_keyed().sum() / (B * (T + 3)) * 100:.0f} %{extra}", flush=True)
    assert not f.any(), "a capacity flag is set: the rate above would be of a run that dropped orders"


results = {}
# ---- host-env
env = env_new(False)
t_sub = t_step = 0.0
for s in range(T + 3):
    t0 = time.perf_counter()
    env.submit_instructions_all(off, batches[s])
    t1 = time.perf_counter()
    env.step()
    t2 = time.perf_counter()
    if s >= 3:
        t_sub += t1 - t0; t_step += t2 - t1
report("host-env", env, t_sub + t_step, f" [submit {t_sub / T * 1e3:.2f} + step {t_step / T * 1e3:.2f} ms]")
ref_l2, ref_tc = env.level2(), env.trade_counts()
env.close()

# ---- sync
env = env_new(True)
for s in range(T + 3):
    if s == 3:
        env.sync(); t0 = time.perf_counter()
    ids = env.submit_instructions_all(off, batches[s])
    env.step(sync=False)
env.sync()
report("sync", env, time.perf_counter() - t0)
assert np.array_equal(env.level2(), ref_l2) and np.array_equal(env.trade_counts(), ref_tc), "sync != host-env"
env.close()

# ---- tickets
env = env_new(True)
prev = None
for s in range(T + 3):
    if s == 3:
        env.sync(); t0 = time.perf_counter()
    t = env.submit_instructions_all_async(off, batches[s])
    env.step(sync=False)
    if prev is not None:
        ids, st, bad = env.submit_result(prev, out=IDS, status=ST)
    prev = t
ids, st, bad = env.submit_result(prev, out=IDS, status=ST)
env.sync()
report("tickets", env, time.perf_counter() - t0)
assert np.array_equal(env.level2(), ref_l2) and np.array_equal(env.trade_counts(), ref_tc), "tickets != host-env"
env.close()

# ---- tickets, results as views of the pinned staging (no copy out)
env = env_new(True)
prev = None
for s in range(T + 3):
    if s == 3:
        env.sync(); t0 = time.perf_counter()
    t = env.submit_instructions_all_async(off, batches[s])
    env.step(sync=False)
    if prev is not None:
        ids, st, bad = env.submit_result(prev, view=True)
    prev = t
ids, st, bad = env.submit_result(prev, view=True)
env.sync()
report("views", env, time.perf_counter() - t0)
assert np.array_equal(env.level2(), ref_l2) and np.array_equal(env.trade_counts(), ref_tc), "views != host-env"
env.close()

# ---- staging: the agent layer writes into the pinned arrays (here: a copy out of the pre-generated batch = the agent's
# own store traffic, timed; a real agent computes straight into them)
env = env_new(True)
prev = None
t_fill = 0.0
for s in range(T + 3):
    if s == 3:
        env.sync(); t0 = time.perf_counter(); t_fill = 0.0
    tf = time.perf_counter()
    stg = env.ingress_staging(n)
    stg["book_offsets"][:] = off
    for k, a in zip(KEYS, batches[s]):
        np.copyto(stg[k][:n], a)
    t_fill += time.perf_counter() - tf
    t = env.submit_instructions_all_async(stg["book_offsets"], tuple(stg[k][:n] for k in KEYS))
    env.step(sync=False)
    if prev is not None:
        ids, st, bad = env.submit_result(prev, out=IDS, status=ST)
    prev = t
ids, st, bad = env.submit_result(prev, out=IDS, status=ST)
env.sync()
report("staging", env, time.perf_counter() - t0, f" [of which the agent's own fill of the staging arrays: {t_fill / T * 1e3:.3f} ms/step]")
assert np.array_equal(env.level2(), ref_l2) and np.array_equal(env.trade_counts(), ref_tc), "staging != host-env"
env.profile(1)
for s in range(5):
    env.submit_instructions_all(off, batches[s])
    env.step(sync=False)
env.sync()
ms, nl = env.profile_read_kind(3)
print(f"k_step_events: {ms / max(nl, 1):.3f} ms per launch ({nl} launches), pool registers R = {env._L.bk_state_bytes_per_book(env._h) // 4 // 320}")
env.close()
