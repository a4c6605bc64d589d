"""ctypes binding of the C ABI declared in include/bourse_amd.h.

The HIP extension is the only execution path: if libbourse_amd.so is missing this raises,
and if no GPU is usable every env-creating call raises ``NoDeviceError``.  There is no CPU
fallback (the CPU oracle under oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os

import numpy as np

from . import _build

BK_OK, BK_PRICE, BK_UNKNOWN_ORDER, BK_CAPACITY, BK_STEP_SIZE, BK_INVALID, BK_HIP, BK_NO_DEVICE = range(8)

FLAG_POOL_OVERFLOW, FLAG_TRADE_OVERFLOW, FLAG_STEP_SIZE, FLAG_ORDER_LOG_FULL = 1, 2, 4, 8
FLAG_UNKNOWN_ORDER, FLAG_HIST_OVERFLOW, FLAG_PRICE_TICK, FLAG_EVENT_OVERFLOW = 16, 32, 64, 128
FLAG_DECODE_LOOKAHEAD = 256
ACTION_MODIFY = 0x80000003  # BK_ACTION_MODIFY: the one extension of submit_instructions' action codes (0 / 1 / 2)
FLAG_NAMES = {1: "POOL_OVERFLOW (live-order pool full: a resting order was dropped)",
              2: "TRADE_OVERFLOW (trade_capacity exceeded: records dropped, counts exact)",
              4: "STEP_SIZE (a step queued >= step_size events)",
              8: "ORDER_LOG_FULL (order id beyond max_orders)",
              16: "UNKNOWN_ORDER", 32: "HIST_OVERFLOW",
              64: "PRICE_TICK (a Noise/Momentum limit price clamped to u32::MAX was not a tick multiple)",
              128: "EVENT_OVERFLOW (a market queued more events in one step than its shared list holds)",
              256: "DECODE_LOOKAHEAD (a ziggurat rejection loop outran the members' decode's 128-draw look-ahead)"}
CAPACITY_FLAGS = FLAG_POOL_OVERFLOW | FLAG_TRADE_OVERFLOW | FLAG_ORDER_LOG_FULL | FLAG_EVENT_OVERFLOW


class BourseError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bourse_amd status {code}: {msg}")
        self.code = code


class NoDeviceError(BourseError):
    pass


class CapacityError(BourseError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("n_books", C.c_uint32), ("levels", C.c_uint32), ("start_time", C.c_uint64),
        ("tick_size", C.c_uint32), ("trading", C.c_uint32), ("step_size", C.c_uint64),
        ("seed", C.c_uint64), ("book_offset", C.c_uint64), ("max_live_orders", C.c_uint32),
        ("max_orders", C.c_uint32), ("trade_capacity", C.c_uint32), ("history_capacity", C.c_uint32),
        ("device", C.c_int32), ("assets", C.c_uint32),
    ]


class RandomAgentsCfg(C.Structure):
    _fields_ = [
        ("n_agents", C.c_uint32), ("tick_lo", C.c_uint32), ("tick_hi", C.c_uint32),
        ("vol_lo", C.c_uint32), ("vol_hi", C.c_uint32), ("tick_size", C.c_uint32),
        ("activity_rate", C.c_float),
    ]


class AgentDesc(C.Structure):
    _fields_ = [
        ("type", C.c_uint32), ("n_agents", C.c_uint32), ("tick_lo", C.c_uint32), ("tick_hi", C.c_uint32),
        ("vol_lo", C.c_uint32), ("vol_hi", C.c_uint32), ("tick_size", C.c_uint32), ("activity_rate", C.c_float),
        ("agent_id_start", C.c_uint32), ("p_limit", C.c_float), ("p_market", C.c_float), ("p_cancel", C.c_float),
        ("trade_vol", C.c_uint32), ("reserved", C.c_uint32), ("price_dist_mu", C.c_double),
        ("price_dist_sigma", C.c_double), ("decay", C.c_double), ("demand", C.c_double), ("scale", C.c_double),
        ("order_ratio", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_books", C.c_uint64), ("sum_trade_vol", C.c_uint64), ("sum_trades", C.c_uint64),
        ("sum_events", C.c_uint64), ("sum_bid_vol", C.c_uint64), ("sum_ask_vol", C.c_uint64),
        ("min_bid", C.c_uint32), ("max_bid", C.c_uint32), ("min_ask", C.c_uint32), ("max_ask", C.c_uint32),
    ]


TRADE_DTYPE = np.dtype(
    {"names": ["t", "side", "price", "vol", "active_id", "passive_id"],
     "formats": ["<u8", "<u4", "<u4", "<u4", "<u8", "<u8"], "offsets": [0, 8, 12, 16, 24, 32], "itemsize": 40})
ORDER_DTYPE = np.dtype(
    {"names": ["side", "status", "arr_time", "end_time", "vol", "start_vol", "price", "trader_id", "order_id"],
     "formats": ["u1", "u1", "<u8", "<u8", "<u4", "<u4", "<u4", "<u4", "<u8"],
     "offsets": [0, 1, 8, 16, 24, 28, 32, 36, 40], "itemsize": 48})

class IngressArrays(C.Structure):
    """bk_ingress_arrays: the pinned staging arrays of the next bk_submit_instructions_host call."""
    _fields_ = [("capacity", C.c_uint64), ("book_offsets", C.c_void_p), ("action", C.c_void_p), ("side", C.c_void_p),
                ("vol", C.c_void_p), ("trader_id", C.c_void_p), ("price", C.c_void_p), ("order_id", C.c_void_p)]


# every symbol include/bourse_amd.h declares: name -> (restype, argtypes)
_u64, _u32, _i32, _vp, _sz = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t
_p64, _p32, _p8 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
SIGNATURES = {
    "bk_last_error": (C.c_char_p, []),
    "bk_hip_versions": (_i32, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bk_device_count": (_i32, [C.POINTER(C.c_int)]),
    "bk_env_create": (_i32, [C.POINTER(Config), C.POINTER(_vp)]),
    "bk_env_destroy": (None, [_vp]),
    "bk_env_set_stream": (_i32, [_vp, _vp]),
    "bk_env_sync": (_i32, [_vp]),
    "bk_place_order": (_i32, [_vp, _u32, _i32, _u32, _u32, _i32, _u32, _p64]),
    "bk_cancel_order": (_i32, [_vp, _u32, _u64]),
    "bk_modify_order": (_i32, [_vp, _u32, _u64, _i32, _u32, _i32, _u32]),
    "bk_submit_instructions": (_i32, [_vp, _u32, _sz, _p32, _p8, _p32, _p32, _p32, _p64, _p64, C.POINTER(_sz)]),
    "bk_submit_instructions_csr": (_i32, [_vp, _p64, _p32, _p8, _p32, _p32, _p32, _p64, _p64, C.POINTER(_sz)]),
    "bk_enable_trading": (_i32, [_vp, _i32]),
    "bk_step": (_i32, [_vp]),
    "bk_device_ingress_enable": (_i32, [_vp, _u32]),
    "bk_submit_instructions_device": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bk_step_async": (_i32, [_vp]),
    "bk_ingress_staging": (_i32, [_vp, _u64, C.POINTER(IngressArrays)]),
    "bk_submit_instructions_host": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _p64]),
    "bk_submit_result": (_i32, [_vp, _u64, _vp, _vp, _p32]),
    "bk_submit_result_view": (_i32, [_vp, _u64, C.POINTER(_vp), C.POINTER(_vp), _p32]),
    "bk_order_status": (_i32, [_vp, _u32, _u64, _p8]),
    "bk_order_count": (_i32, [_vp, _u32, _p64]),
    "bk_get_orders": (_i32, [_vp, _u32, _u64, _u64, _vp]),
    "bk_get_order_keys": (_i32, [_vp, _u32, _u64, _u64, _p32, _p64]),
    "bk_load_book": (_i32, [_vp, _u32, _u64, _u32, _u64, _vp, _p32, _p64, _u64, _vp]),
    "bk_set_random_agents": (_i32, [_vp, _u32, C.POINTER(RandomAgentsCfg)]),
    "bk_set_tick_sizes": (_i32, [_vp, _u32, _p32]),
    "bk_set_random_market_agents": (_i32, [_vp, _u32, C.POINTER(RandomAgentsCfg), _p32]),
    "bk_set_agents": (_i32, [_vp, _u32, C.POINTER(AgentDesc)]),
    "bk_set_market_agents": (_i32, [_vp, _u32, C.POINTER(AgentDesc), _p32]),
    "bk_run": (_i32, [_vp, _u64]),
    "bk_l2_width": (_u32, [_vp]),
    "bk_level2": (_i32, [_vp, _u32, _u32, _p32]),
    "bk_history_len": (_i32, [_vp, _p64, _p64]),
    "bk_history": (_i32, [_vp, _u64, _u64, _u32, _u32, _p32]),
    "bk_clear_history": (_i32, [_vp]),
    "bk_history_copy_async": (_i32, [_vp, _u64, _u64, _u32, _u32, _p32, _vp]),
    "bk_stream_create": (_i32, [C.POINTER(_vp)]),
    "bk_stream_sync": (_i32, [_vp]),
    "bk_stream_destroy": (_i32, [_vp]),
    "bk_pinned_alloc": (_i32, [_u64, C.POINTER(_vp)]),
    "bk_pinned_free": (_i32, [_vp]),
    "bk_trade_count": (_i32, [_vp, _u32, _p64, _p64]),
    "bk_trade_counts": (_i32, [_vp, _p64]),
    "bk_get_trades": (_i32, [_vp, _u32, _u64, _u64, _vp]),
    "bk_clear_trades": (_i32, [_vp]),
    "bk_trades_compact": (_i32, [_vp, _p64]),
    "bk_trades_compact_copy_async": (_i32, [_vp, _vp, _p64, _vp]),
    "bk_time": (_i32, [_vp, _u32, _p64]),
    "bk_set_time": (_i32, [_vp, _u32, _u64]),
    "bk_trade_vol": (_i32, [_vp, _u32, _p32]),
    "bk_steps_done": (_i32, [_vp, _p64]),
    "bk_book_flags": (_i32, [_vp, _p32]),
    "bk_clear_flags": (_i32, [_vp, _u32]),
    "bk_flags_summary": (_i32, [_vp, _p32, _p64]),
    "bk_warm": (_i32, [_vp, _u64]),
    "bk_rng_state": (_i32, [_vp, _u32, _p64]),
    "bk_live_orders": (_i32, [_vp, _u32, _u32, _vp, _p32]),
    "bk_stats_compute": (_i32, [_vp, C.POINTER(Stats)]),
    "bk_stats_device_ptr": (_i32, [_vp, C.POINTER(_vp)]),
    "bk_level2_device_ptr": (_i32, [_vp, C.POINTER(_vp)]),
    "bk_profile_enable": (_i32, [_vp, _i32]),
    "bk_profile_read": (_i32, [_vp, C.POINTER(C.c_double), _p64, _i32]),
    "bk_profile_read_kind": (_i32, [_vp, _i32, C.POINTER(C.c_double), _p64]),
    "bk_set_pipeline": (_i32, [_vp, _i32]),
    "bk_get_pipeline": (_i32, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bk_state_bytes_per_book": (_u64, [_vp]),
    "bk_set_split_parts": (_i32, [_vp, _i32, _u32]),
    "bk_get_split_parts": (_i32, [_vp, C.POINTER(C.c_int), _p32]),
    "bk_set_wave_options": (_i32, [_vp, _u32, _i32]),
    "bk_pipeline_fallbacks": (_i32, [_vp, _p64]),
    "bk_order_counts": (_i32, [_vp, _p64]),
    "bk_event_steps_keyed": (_i32, [_vp, _p64]),
    "bk_checkpoint_bytes": (_u64, [_vp]),
    "bk_checkpoint_save": (_i32, [_vp, _vp, _u64]),
    "bk_checkpoint_load": (_i32, [_vp, _vp, _u64]),
}

_lib = None


def lib_path() -> str:
    return _build.LIB


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheels bundle their own libamdhip64.so.7 - the SONAME of /opt/rocm's, which this library links.  A process
    gets ONE of the two (the first loaded), and torch does not find the GPU on the system's copy ("No HIP GPUs are available"
    when `import torch` comes AFTER this library's first use; the other order has always worked, and is what bench.py and the
    tests do).  So when torch is installed but not imported yet, its copy of the runtime is loaded first - without importing
    torch.  BOURSE_AMD_OWN_HIP_RUNTIME=1 skips this (a process that never imports torch then runs on /opt/rocm's runtime, as a
    plain C client of the library does)."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("BOURSE_AMD_OWN_HIP_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            _note(f"HIP runtime: torch's bundled copy pre-loaded ({cand}); BOURSE_AMD_OWN_HIP_RUNTIME=1 keeps /opt/rocm's")
        except OSError as e:
            # (not loadable here: the library's own dependency resolves as before)
            _note(f"HIP runtime: torch's bundled copy could not be loaded ({e}); the library binds to its own dependency")


def _note(msg: str):
    if os.environ.get("BOURSE_AMD_VERBOSE"):
        import sys

        print("[bourse_amd] " + msg, file=sys.stderr)


def _check_runtime_version(L):
    """Which HIP runtime did the library bind to, and is it the major version it was built against?  (the
    pre-load above silently changes the runtime of every process that has torch installed.)  A mismatch of the major
    version warns; BOURSE_AMD_VERBOSE=1 prints both versions and the runtime's path."""
    try:
        built, run = C.c_int(0), C.c_int(0)
        if L.bk_hip_versions(C.byref(built), C.byref(run)) != 0:
            return
        path = "?"
        try:
            for line in open("/proc/self/maps"):
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        except OSError:
            pass
        _note(f"HIP runtime bound: version {run.value} at {path}; library built against {built.value}")
        if built.value // 10_000_000 != run.value // 10_000_000:
            import warnings

            warnings.warn(f"bourse_amd: built against HIP {built.value} but running on HIP runtime {run.value} ({path}); set "
                          f"BOURSE_AMD_OWN_HIP_RUNTIME=1 to keep the system runtime", RuntimeWarning)
    except Exception:  # noqa: BLE001  (diagnostics must never stop a load)
        pass


def load() -> C.CDLL:
    """Load libbourse_amd.so (building it first if hipcc is here and the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    path = _build.LIB
    override = os.environ.get("BOURSE_AMD_LIBRARY")  # a variant build (scripts/asm_ab.sh); never rebuilt implicitly
    if override:
        if not os.path.exists(override):
            raise ImportError(f"bourse_amd: BOURSE_AMD_LIBRARY={override} does not exist")
        path = override
    elif _build.is_stale():
        try:
            _build.build()
        except Exception as e:  # no toolchain on this machine
            if not os.path.exists(path):
                raise ImportError(
                    f"bourse_amd: HIP extension {path} is missing and could not be built ({e}). "
                    "There is no CPU fallback."
                ) from e
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    L.bk_selftest_reduce.restype = _i32
    L.bk_selftest_reduce.argtypes = [_p32, _u32, _p32]
    L.bk_selftest_math.restype = _i32
    L.bk_selftest_math.argtypes = [_i32, C.POINTER(C.c_double), _u64, C.POINTER(C.c_double)]
    _lib = L
    _check_runtime_version(L)
    return L


def check(rc: int):
    if rc == BK_OK:
        return
    msg = load().bk_last_error().decode()
    if rc == BK_PRICE:
        raise ValueError(msg)  # reference: PyValueError(e.to_string()), rust/src/step_sim.rs:239-242
    if rc == BK_NO_DEVICE:
        raise NoDeviceError(rc, msg)
    if rc == BK_CAPACITY:
        raise CapacityError(rc, msg)
    if rc == BK_UNKNOWN_ORDER:
        raise IndexError(msg)
    raise BourseError(rc, msg)


def p32(a):
    return a.ctypes.data_as(_p32)


def p64(a):
    return a.ctypes.data_as(_p64)


def p8(a):
    return a.ctypes.data_as(_p8)
