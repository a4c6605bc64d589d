"""Where does a wave's time go UNDER THE REAL PIPELINE'S LOAD?  Needs the diagnostic build (-DBOURSE_AMD_STAMPS=1:
build_variants/lib_stamps.so, `BOURSE_AMD_LIBRARY=...`): every k_step_batch / k_agents_wave wave adds the shader-clock
length of its phases to a device array.   GPU box:  BOURSE_AMD_LIBRARY=build_variants/lib_stamps.so python scripts/wave_phases.py [books] [pipeline]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bourse_amd as bk
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pipe = sys.argv[2] if len(sys.argv) > 2 else "auto"
WL = sys.argv[3] if len(sys.argv) > 3 else "C3"
SKEW = "--skew" in sys.argv  # also: one step on its own, every wave's absolute start / end inside its launch
if WL == "C5":
    G, LV, POOL = [(256, (100, 164), (10, 20), 2, 0.8), (256, (100, 164), (50, 70), 2, 0.2)], 64, 512
elif WL == "C2":
    G, LV, POOL = [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)], 16, 64
else:
    G, LV, POOL = [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)], 32, 128
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=LV, max_live_orders=POOL, trade_capacity=POOL // 2 * 50, history_capacity=50, strict=False)
env.set_random_agents(G)
env.set_pipeline(pipe)
L = env._L
L.bk_debug_stamps.argtypes = [C.c_uint32, C.c_void_p]
W = 24  # BK_STAMP_WORDS
buf = np.zeros((B, W), dtype=np.uint32)
L.bk_debug_stamps(B, None)  # allocate + zero BEFORE the first launch (the kernels write through the pointer)
env.run(50); env.clear_history(); env.clear_trades()
L.bk_debug_stamps(B, buf.ctypes.data_as(C.c_void_p))  # (reads the warm-up's stamps and zeroes)
t0 = time.perf_counter()
T = 100
for _ in range(T // 50):
    env.run(50, sync=False); env.clear_history(); env.clear_trades()
env.sync()
dt = time.perf_counter() - t0
L.bk_debug_stamps(B, buf.ctypes.data_as(C.c_void_p))
a = buf.astype(np.float64)
a[:, [2, 6, 12, 13]] = 0  # (the absolute stamps of --skew)
a = a.sum(axis=0).reshape(3, 8)
print(f"{B} books, pipeline {env.pipeline()}: {B * T / dt / 1e6:.1f} M book-steps/s with the stamps in (s_memtime per phase)")
names = {0: ("k_step_batch", ["loads' round trip", "unpack + masks + new orders", "-", "keys + event loop", "snapshot + trade flush", "store"]),
         1: ("k_agents_wave", ["lane-state cache in", "agents.update: group loop, the rest", "shuffle: resolution", "publish"]),
         2: ("k_agents_wave, inside", ["generation", "window: masks, searches, continuations", "walk", "window's events out", "shuffle: draws + acceptance"])}
a[2, 7] = a[1, 7]
for k, (kn, ph) in names.items():
    n = a[k, 7]
    if not n:
        continue
    tot = a[k, :7].sum()
    print(f"  {kn}: {int(n)} waves, {tot / n:.0f} clocks per wave = {tot / n / 2.4e3:.1f} us at 2.4 GHz (s_memtime ticks; if 100 MHz: x24)")
    for i, p in enumerate(ph):
        if p != "-":
            print(f"     {p:42s} {a[k, i] / n:9.0f} clocks  {100 * a[k, i] / tot:5.1f} %")

def skew(title):
    parts = max(1, env.pipeline()[1])
    print(title)
    for k, (kn, st, en) in {0: ("k_step_batch", 6, 2), 1: ("k_agents_wave", 12, 13)}.items():
        if not buf[:, k * 8 + 7].any():
            continue
        per = (B + parts - 1) // parts
        for pi in range(parts):
            sl = slice(pi * per, min(B, (pi + 1) * per))
            s0 = buf[sl, st].astype(np.int64)
            e0 = buf[sl, en].astype(np.int64)
            base = s0.min()
            s_rel = (s0 - base) & 0xFFFFFFFF
            e_rel = (e0 - base) & 0xFFFFFFFF
            d = (e0 - s0) & 0xFFFFFFFF
            q = lambda x, p: float(np.percentile(x, p)) / 2.24e3
            print(f"  {kn} part {pi} ({sl.stop - sl.start} waves), us at 2.24 GHz: wave life p50 {q(d, 50):.1f} p90 {q(d, 90):.1f} max {q(d, 100):.1f};"
                  f" start p50 {q(s_rel, 50):.1f} p90 {q(s_rel, 90):.1f} max {q(s_rel, 100):.1f}; end p50 {q(e_rel, 50):.1f} p99 {q(e_rel, 99):.1f} max {q(e_rel, 100):.1f}")


if SKEW:
    skew("last step of the run above (all parts in flight):")
    env.sync()
    env.run(1)
    env.sync()
    L.bk_debug_stamps(B, buf.ctypes.data_as(C.c_void_p))
    skew("one step alone on the device:")
