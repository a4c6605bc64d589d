"""How often C5 as written leaves the keyed loop's price window, why, and how often the top-anchored window (keys_begin_wide)
takes the step instead.  Needs the diagnostic build: BOURSE_AMD_LIBRARY=build_variants/lib_stamps.so python scripts/keyed_tally.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bourse_amd as bk
sys.path.insert(0, ROOT)
import bench
B, T = 8192, 100
levels, members = bench.WORKLOADS["C5M"][1], bench.WORKLOADS["C5M"][2]
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, levels=levels, max_live_orders=512, trade_capacity=384 * 50, history_capacity=50, strict=False)
env.set_agents(members)
L = env._L
L.bk_debug_stamps.argtypes = [C.c_uint32, C.c_void_p]
buf = np.zeros((B, 24), dtype=np.uint32)
L.bk_debug_stamps(B, None)
env.run(50); env.clear_history(); env.clear_trades()
L.bk_debug_stamps(B, buf.ctypes.data_as(C.c_void_p))
for _ in range(T // 50):
    env.run(50); env.clear_history(); env.clear_trades()
L.bk_debug_stamps(B, buf.ctypes.data_as(C.c_void_p))
a = buf.astype(np.int64).sum(axis=0)
n = a[7]
print(f"C5 as written, {B} books x {T} steps, pipeline {env.pipeline()}: {n} book-steps of k_step_batch")
print(f"  narrow window failed: {a[9]} ({100 * a[9] / n:.1f} %); of those: price at u32::MAX {a[10]}, arrival span {a[11]}, "
      f"volume / no price {a[12]}, an ask below the window {a[13]}, liquidity guard {a[14]}, taken by the wide window {a[15]} "
      f"({100 * a[15] / max(a[9], 1):.1f} %); still on the two-reduction loop: {100 * (a[9] - a[15]) / n:.1f} % of the book-steps")
per_book = buf[:, 9].astype(np.int64) - buf[:, 15]
print(f"  books with at least one two-reduction step in {T}: {(per_book > 0).sum()} of {B}")
