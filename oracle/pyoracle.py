"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``bourse_amd/`` does.

The classes mirror the reference's PyO3 classes so the oracle can stand in for
``bourse.core`` when the reference's own Python callers/tests are exercised in
the build container (see ``oracle/run_reference_pytests.py``):

* ``StepEnv``        -- ref rust/src/step_sim.rs:55-607
* ``StepEnvNumpy``   -- ref rust/src/step_sim_numpy.rs:66-516
* ``OrderBook``      -- ref rust/src/order_book.rs:30-398
* ``RandomAgents`` / ``sim_runner`` -- ref crates/step_sim/src/agents/random_agent.rs,
  crates/step_sim/src/runner.rs:46-69
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbourse_oracle.so")
_SOA_PATH = os.path.join(_HERE, "libbourse_soa.so")

U64_MAX = 2**64 - 1
MAX_PRICE = 2**32 - 1

ORDER_DTYPE = np.dtype(
    {
        "names": ["side", "status", "arr_time", "end_time", "vol", "start_vol", "price", "trader_id", "order_id"],
        "formats": ["u1", "u1", "<u8", "<u8", "<u4", "<u4", "<u4", "<u4", "<u8"],
        "offsets": [0, 1, 8, 16, 24, 28, 32, 36, 40],
        "itemsize": 48,
    }
)
TRADE_DTYPE = np.dtype(
    {
        "names": ["t", "side", "price", "vol", "active_id", "passive_id"],
        "formats": ["<u8", "<u4", "<u4", "<u4", "<u8", "<u8"],
        "offsets": [0, 8, 12, 16, 24, 32],
        "itemsize": 40,
    }
)


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (seconds).  Returns the library path."""
    srcs = [os.path.join(_HERE, f) for f in ("bourse_oracle.cpp", "bourse_oracle_capi.cpp", "bourse_oracle.hpp",
                                             "bourse_oracle_agents.cpp", "bourse_oracle_agents.hpp", "pm_math.hpp",
                                             "zig_norm_tables.inc")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "libbourse_oracle.so"], check=True, capture_output=True)
    soa_src = os.path.join(_HERE, "bourse_soa.cpp")
    if force or not os.path.exists(_SOA_PATH) or os.path.getmtime(soa_src) > os.path.getmtime(_SOA_PATH):
        subprocess.run(["make", "-C", _HERE, "libbourse_soa.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u64, u32, i32, vp, f32 = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_float
    p64, p32, p8 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("orc_many_set_build_threads", None, i32)
    sig("orc_rng_seed", None, u64, p64)
    sig("orc_rng_next_u64", u64, p64)
    sig("orc_rng_next_u32", u32, p64)
    sig("orc_rng_f32", f32, p64)
    sig("orc_rng_range", u32, p64, u32, u32)
    sig("orc_rng_shuffle_u32", None, p64, p32, u64)

    sig("orc_side_new", vp, i32)
    sig("orc_side_free", None, vp)
    sig("orc_side_insert", None, vp, u64, u32, u64, u32)
    sig("orc_side_remove_order", None, vp, u64, u32, u32)
    sig("orc_side_remove_vol", None, vp, u32, u32)
    sig("orc_side_vol", u32, vp)
    sig("orc_side_best_price", u32, vp)
    sig("orc_side_best_vol_and_orders", None, vp, p32)
    sig("orc_side_best_order_idx", i32, vp, p64)
    sig("orc_side_vol_and_orders_at_price", None, vp, u32, p32)

    sig("orc_book_new", vp, u64, u32, i32, i32)
    sig("orc_book_free", None, vp)
    sig("orc_book_set_time", None, vp, u64)
    sig("orc_book_get_time", u64, vp)
    sig("orc_book_set_trading", None, vp, i32)
    sig("orc_book_create_order", i32, vp, i32, u32, u32, i32, u32, p64)
    sig("orc_book_place_order", i32, vp, u64)
    sig("orc_book_create_and_place", i32, vp, i32, u32, u32, i32, u32, p64)
    sig("orc_book_cancel", i32, vp, u64)
    sig("orc_book_modify", i32, vp, u64, i32, u32, i32, u32)
    sig("orc_book_bid_ask", None, vp, p32)
    sig("orc_book_bid_vol", u32, vp)
    sig("orc_book_ask_vol", u32, vp)
    sig("orc_book_trade_vol", u32, vp)
    sig("orc_book_best_bid_vol_and_orders", None, vp, p32)
    sig("orc_book_best_ask_vol_and_orders", None, vp, p32)
    sig("orc_book_mid_price", C.c_double, vp)
    sig("orc_book_level2", None, vp, p32)
    sig("orc_book_n_orders", u64, vp)
    sig("orc_book_n_trades", u64, vp)
    sig("orc_book_order_status", i32, vp, u64, p8)
    sig("orc_book_get_orders", None, vp, vp, u64, u64)
    sig("orc_book_get_trades", None, vp, vp, u64, u64)
    sig("orc_book_get_keys", None, vp, u64, u64, p8, p32, p64)
    sig("orc_book_from_state", vp, u64, u32, u32, i32, i32, u64, vp, p8, p32, p64, u64, vp)

    sig("orc_env_new", vp, u64, u64, u32, u64, i32, i32)
    sig("orc_env_free", None, vp)
    sig("orc_env_book", vp, vp)
    sig("orc_env_rng_state", None, vp, p64)
    sig("orc_env_place_order", i32, vp, i32, u32, u32, i32, u32, p64)
    sig("orc_env_cancel_order", None, vp, u64)
    sig("orc_env_modify_order", None, vp, u64, i32, u32, i32, u32)
    sig("orc_env_n_transactions", u64, vp)
    sig("orc_env_transaction_kinds", None, vp, p8)
    sig("orc_env_step", i32, vp)
    sig("orc_env_level2", None, vp, p32)
    sig("orc_env_n_steps", u64, vp)
    sig("orc_env_history", None, vp, p32)

    sig("orc_agents_new", vp)
    sig("orc_agents_free", None, vp)
    sig("orc_agents_add_random", None, vp, u64, u32, u32, u32, u32, u32, f32)
    sig("orc_agents_held_ids", None, vp, i32, p64)
    sig("orc_agents_update", None, vp, vp)
    sig("orc_sim_run", i32, vp, vp, p64, u64)

    sig("orc_many_new", vp, u32, u64, u64, u32, u64, i32, i32, i32, p32)
    sig("orc_many_free", None, vp)
    sig("orc_many_run", i32, vp, u64, i32)
    sig("orc_many_n_steps", u64, vp)
    sig("orc_many_book", vp, vp, u32)
    sig("orc_many_rng_state", None, vp, u32, p64)
    sig("orc_many_history", None, vp, u64, u64, p32)
    sig("orc_many_trade_counts", None, vp, p64)
    sig("orc_many_order_counts", None, vp, p64)
    dbl = C.c_double
    sig("orc_agents_add_desc", None, vp, vp)
    sig("orc_agents_order_list", u64, vp, i32, p64, u64)
    sig("orc_agents_set_noise_prob", None, vp, i32, i32, f32)
    sig("orc_many_new_mixed", vp, u32, u64, u64, u32, u64, i32, i32, i32, vp)
    sig("orc_rng_f64", dbl, p64)
    sig("orc_rng_std_normal", dbl, p64)
    sig("orc_rng_lognormal", dbl, p64, dbl, dbl)
    sig("orc_pm_exp", dbl, dbl)
    sig("orc_pm_log", dbl, dbl)
    sig("orc_pm_tanh", dbl, dbl)
    sig("orc_round_price_up", u32, dbl, dbl)
    sig("orc_round_price_down", u32, dbl, dbl)
    sig("orc_version", i32)
    sig("orc_mkts_new", vp, u32, u64, u64, u32, p32, u64, i32, i32, i32, p32)
    sig("orc_mkts_new_mixed", vp, u32, u64, u64, u32, p32, u64, i32, i32, i32, vp, p32)
    sig("orc_mkts_free", None, vp)
    sig("orc_mkts_run", i32, vp, u64, i32)
    sig("orc_mkts_place", i32, vp, u32, u32, i32, u32, u32, i32, u32, p64)
    sig("orc_mkts_cancel", None, vp, u32, u32, u64)
    sig("orc_mkts_modify", None, vp, u32, u32, u64, i32, u32, i32, u32)
    sig("orc_mkts_step", i32, vp)
    sig("orc_mkts_set_trading", None, vp, i32)
    sig("orc_mkts_n_steps", u64, vp)
    sig("orc_mkts_book", vp, vp, u32, u32)
    sig("orc_mkts_rng_state", None, vp, u32, p64)
    sig("orc_mkts_history", None, vp, u64, u64, p32)
    _lib = L
    return L


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _price_error(price, tick):
    # ref orderbook.rs:135-139 Display text; surfaced as ValueError (step_sim.rs:239-242)
    return ValueError(f"Price {price} was not a multiple of tick-size {tick}")


# ---------------------------------------------------------------------------- RNG
class Rng:
    """xoroshiro128** stream as the reference seeds it (seed_from_u64)."""

    def __init__(self, seed=None, state=None):
        self.st = np.zeros(2, dtype=np.uint64)
        if state is not None:
            self.st[:] = state
        else:
            lib().orc_rng_seed(int(seed), _p64(self.st))

    def next_u64(self):
        return int(lib().orc_rng_next_u64(_p64(self.st)))

    def next_u32(self):
        return int(lib().orc_rng_next_u32(_p64(self.st)))

    def gen_f32(self):
        return np.float32(lib().orc_rng_f32(_p64(self.st)))

    def gen_range(self, lo, hi):
        return int(lib().orc_rng_range(_p64(self.st), lo, hi))

    def gen_f64(self):
        return float(lib().orc_rng_f64(_p64(self.st)))

    def std_normal(self):
        return float(lib().orc_rng_std_normal(_p64(self.st)))

    def lognormal(self, mu, sigma):
        return float(lib().orc_rng_lognormal(_p64(self.st), mu, sigma))

    def shuffle(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint32)
        lib().orc_rng_shuffle_u32(_p64(self.st), _p32(a), len(a))
        return a


# ------------------------------------------------------------------------ side
class BookSide:
    """One side of the book (ref crates/order_book/src/side.rs); kind in {"raw","ask","bid"}."""

    def __init__(self, kind="raw"):
        self._s = lib().orc_side_new({"raw": 0, "ask": 1, "bid": 2}[kind])

    def __del__(self):
        if getattr(self, "_s", None):
            lib().orc_side_free(self._s)
            self._s = None

    def insert_order(self, t, price, idx, vol):
        lib().orc_side_insert(self._s, int(t), int(price), int(idx), int(vol))

    def remove_order(self, t, price, vol):
        lib().orc_side_remove_order(self._s, int(t), int(price), int(vol))

    def remove_vol(self, price, vol):
        lib().orc_side_remove_vol(self._s, int(price), int(vol))

    def vol(self):
        return int(lib().orc_side_vol(self._s))

    def best_price(self):
        return int(lib().orc_side_best_price(self._s))

    def best_vol_and_orders(self):
        o = np.zeros(2, dtype=np.uint32)
        lib().orc_side_best_vol_and_orders(self._s, _p32(o))
        return int(o[0]), int(o[1])

    def best_vol(self):
        return self.best_vol_and_orders()[0]

    def best_order_idx(self):
        out = C.c_uint64(0)
        return int(out.value) if lib().orc_side_best_order_idx(self._s, C.byref(out)) else None

    def vol_and_orders_at_price(self, price):
        o = np.zeros(2, dtype=np.uint32)
        lib().orc_side_vol_and_orders_at_price(self._s, int(price), _p32(o))
        return int(o[0]), int(o[1])


# ------------------------------------------------------------------- book views
class _BookView:
    """Queries on an oracle OrderBook pointer (owned elsewhere)."""

    def __init__(self, ptr, levels, tick):
        self._b = ptr
        self._levels = levels
        self._tick = tick

    def bid_ask(self):
        o = np.zeros(2, dtype=np.uint32)
        lib().orc_book_bid_ask(self._b, _p32(o))
        return int(o[0]), int(o[1])

    def bid_vol(self):
        return int(lib().orc_book_bid_vol(self._b))

    def ask_vol(self):
        return int(lib().orc_book_ask_vol(self._b))

    def best_bid_vol_and_orders(self):
        o = np.zeros(2, dtype=np.uint32)
        lib().orc_book_best_bid_vol_and_orders(self._b, _p32(o))
        return int(o[0]), int(o[1])

    def best_ask_vol_and_orders(self):
        o = np.zeros(2, dtype=np.uint32)
        lib().orc_book_best_ask_vol_and_orders(self._b, _p32(o))
        return int(o[0]), int(o[1])

    def best_bid_vol(self):
        return self.best_bid_vol_and_orders()[0]

    def best_ask_vol(self):
        return self.best_ask_vol_and_orders()[0]

    def mid_price(self):
        return float(lib().orc_book_mid_price(self._b))

    def trade_vol(self):
        return int(lib().orc_book_trade_vol(self._b))

    def get_time(self):
        return int(lib().orc_book_get_time(self._b))

    def level2(self):
        """(bid_price, ask_price, bid_vol, ask_vol, bid_levels[L,2], ask_levels[L,2])"""
        L = self._levels
        o = np.zeros(4 + 4 * L, dtype=np.uint32)
        lib().orc_book_level2(self._b, _p32(o))
        return int(o[0]), int(o[1]), int(o[2]), int(o[3]), o[4 : 4 + 2 * L].reshape(L, 2), o[4 + 2 * L :].reshape(L, 2)

    def order_status(self, order_id):
        out = C.c_uint8(0)
        rc = lib().orc_book_order_status(self._b, int(order_id), C.byref(out))
        if rc != 0:
            raise IndexError(f"No order with id {order_id} exists")
        return int(out.value)

    def n_orders(self):
        return int(lib().orc_book_n_orders(self._b))

    def n_trades(self):
        return int(lib().orc_book_n_trades(self._b))

    def orders_array(self):
        n = self.n_orders()
        a = np.zeros(n, dtype=ORDER_DTYPE)
        if n:
            lib().orc_book_get_orders(self._b, a.ctypes.data_as(C.c_void_p), 0, n)
        return a

    def trades_array(self):
        n = self.n_trades()
        a = np.zeros(n, dtype=TRADE_DTYPE)
        if n:
            lib().orc_book_get_trades(self._b, a.ctypes.data_as(C.c_void_p), 0, n)
        return a

    def get_orders(self):
        # PyOrder tuples, ref rust/src/types.rs:17-31
        return [
            (bool(r["side"]), int(r["status"]), int(r["arr_time"]), int(r["end_time"]), int(r["vol"]),
             int(r["start_vol"]), int(r["price"]), int(r["trader_id"]), int(r["order_id"]))
            for r in self.orders_array()
        ]

    def get_trades(self):
        # PyTrade tuples, ref rust/src/types.rs:4-15
        return [
            (int(r["t"]), bool(r["side"]), int(r["price"]), int(r["vol"]), int(r["active_id"]), int(r["passive_id"]))
            for r in self.trades_array()
        ]


class OrderBook(_BookView):
    """Immediate-mode book; mirrors ``bourse.core.OrderBook`` (ref rust/src/order_book.rs)."""

    def __init__(self, start_time, tick_size, trading=True, levels=10):
        ptr = lib().orc_book_new(int(start_time), int(tick_size), int(bool(trading)), int(levels))
        super().__init__(ptr, levels, tick_size)
        self._trading = bool(trading)
        self._own = True

    def __del__(self):
        if getattr(self, "_own", False) and self._b:
            lib().orc_book_free(self._b)
            self._b = None

    def set_time(self, t):
        lib().orc_book_set_time(self._b, int(t))

    def enable_trading(self):
        self._trading = True
        lib().orc_book_set_trading(self._b, 1)

    def disable_trading(self):
        self._trading = False
        lib().orc_book_set_trading(self._b, 0)

    def create_order(self, bid, vol, trader_id, price=None):
        out = C.c_uint64(0)
        rc = lib().orc_book_create_order(self._b, int(bool(bid)), int(vol), int(trader_id), int(price is not None),
                                         int(price or 0), C.byref(out))
        if rc == 1:
            raise _price_error(price, self._tick)
        return int(out.value)

    def place_order_id(self, order_id):
        rc = lib().orc_book_place_order(self._b, int(order_id))
        if rc != 0:
            raise IndexError(order_id)

    def place_order(self, bid, vol, trader_id, price=None):
        out = C.c_uint64(0)
        rc = lib().orc_book_create_and_place(self._b, int(bool(bid)), int(vol), int(trader_id),
                                             int(price is not None), int(price or 0), C.byref(out))
        if rc == 1:
            raise _price_error(price, self._tick)
        return int(out.value)

    def cancel_order(self, order_id):
        rc = lib().orc_book_cancel(self._b, int(order_id))
        if rc != 0:
            raise RuntimeError(f"No order with id {order_id} exists")  # reference: panic!

    def modify_order(self, order_id, new_price=None, new_vol=None):
        rc = lib().orc_book_modify(self._b, int(order_id), int(new_price is not None), int(new_price or 0),
                                   int(new_vol is not None), int(new_vol or 0))
        if rc != 0:
            raise IndexError(order_id)

    # JSON snapshot in the reference's serde layout (orderbook.rs:93-112: t, tick_size, trade_vol, orders [{order,
    # key}], trades, trading; unit enum variants as strings, OrderKey as a 3-array; the sides are not serialised).
    def state(self):
        side = {1: "Bid", 0: "Ask"}
        status = ["New", "Active", "Filled", "Cancelled", "Rejected"]
        o = self.orders_array()
        n = len(o)
        kb, kp, kt = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint64)
        if n:
            lib().orc_book_get_keys(self._b, 0, n, kb.ctypes.data_as(C.POINTER(C.c_uint8)), _p32(kp), _p64(kt))
        orders = [{
            "order": {"side": side[int(r["side"])], "status": status[int(r["status"])], "arr_time": int(r["arr_time"]),
                      "end_time": int(r["end_time"]), "vol": int(r["vol"]), "start_vol": int(r["start_vol"]),
                      "price": int(r["price"]), "trader_id": int(r["trader_id"]), "order_id": int(r["order_id"])},
            "key": [side[int(kb[i])], int(kp[i]), int(kt[i])],
        } for i, r in enumerate(o)]
        trades = [{"t": int(r["t"]), "side": side[int(r["side"])], "price": int(r["price"]), "vol": int(r["vol"]),
                   "active_order_id": int(r["active_id"]), "passive_order_id": int(r["passive_id"])}
                  for r in self.trades_array()]
        return {"t": self.get_time(), "tick_size": self._tick, "trade_vol": self.trade_vol(), "orders": orders,
                "trades": trades, "trading": self._trading}

    def save_json_snapshot(self, path, pretty=False):
        with open(path, "w") as f:
            if pretty:
                json.dump(self.state(), f, indent=2)
            else:
                json.dump(self.state(), f, separators=(",", ":"))


def order_book_from_state(s, levels=10):
    """``TryFrom<OrderBookState>`` (orderbook.rs:891-918)."""
    side = {"Bid": 1, "Ask": 0}
    status = {"New": 0, "Active": 1, "Filled": 2, "Cancelled": 3, "Rejected": 4}
    n = len(s["orders"])
    o = np.zeros(n, dtype=ORDER_DTYPE)
    kb, kp, kt = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint64)
    for i, e in enumerate(s["orders"]):
        r, k = e["order"], e["key"]
        o[i] = (side[r["side"]], status[r["status"]], r["arr_time"], r["end_time"], r["vol"], r["start_vol"],
                r["price"], r["trader_id"], r["order_id"])
        kb[i], kp[i], kt[i] = side[k[0]], k[1], k[2]
    t = np.zeros(len(s["trades"]), dtype=TRADE_DTYPE)
    for i, r in enumerate(s["trades"]):
        t[i] = (r["t"], side[r["side"]], r["price"], r["vol"], r["active_order_id"], r["passive_order_id"])
    ob = OrderBook.__new__(OrderBook)
    ptr = lib().orc_book_from_state(int(s["t"]), int(s["tick_size"]), int(s["trade_vol"]), int(bool(s["trading"])),
                                    int(levels), n, o.ctypes.data_as(C.c_void_p),
                                    kb.ctypes.data_as(C.POINTER(C.c_uint8)), _p32(kp), _p64(kt), len(t),
                                    t.ctypes.data_as(C.c_void_p))
    _BookView.__init__(ob, ptr, levels, int(s["tick_size"]))
    ob._trading, ob._own = bool(s["trading"]), True
    return ob


def order_book_from_json(path):
    with open(path) as f:
        return order_book_from_state(json.load(f))


class _EnvBase:
    LEVELS = 10

    def __init__(self, seed, start_time, tick_size, step_size, trading=True, levels=None):
        self._levels = int(levels if levels is not None else self.LEVELS)
        self._tick = int(tick_size)
        self._e = lib().orc_env_new(int(seed), int(start_time), int(tick_size), int(step_size), int(bool(trading)),
                                    self._levels)
        self._book = _BookView(lib().orc_env_book(self._e), self._levels, self._tick)

    def __del__(self):
        if getattr(self, "_e", None):
            lib().orc_env_free(self._e)
            self._e = None

    @property
    def book(self):
        return self._book

    def rng_state(self):
        st = np.zeros(2, dtype=np.uint64)
        lib().orc_env_rng_state(self._e, _p64(st))
        return st

    def enable_trading(self):
        lib().orc_book_set_trading(self._book._b, 1)

    def disable_trading(self):
        lib().orc_book_set_trading(self._book._b, 0)

    def step(self):
        rc = lib().orc_env_step(self._e)
        if rc == 2:
            raise RuntimeError("No order with that id exists")  # reference: panic! (orderbook.rs:642)

    def _place(self, bid, vol, trader_id, price):
        out = C.c_uint64(0)
        rc = lib().orc_env_place_order(self._e, int(bool(bid)), int(vol), int(trader_id), int(price is not None),
                                       int(price if price is not None else 0), C.byref(out))
        if rc == 1:
            raise _price_error(price, self._tick)
        return int(out.value)

    def n_transactions(self):
        return int(lib().orc_env_n_transactions(self._e))

    def transaction_kinds(self):
        n = self.n_transactions()
        out = np.zeros(n, dtype=np.uint8)
        if n:
            lib().orc_env_transaction_kinds(self._e, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def _l2(self):
        o = np.zeros(5 + 4 * self._levels, dtype=np.uint32)
        lib().orc_env_level2(self._e, _p32(o))
        return o

    def history(self):
        """u32[T, 5+4L] in the numpy level_2_data layout (trade_vol = per-step record)."""
        T = int(lib().orc_env_n_steps(self._e))
        o = np.zeros((T, 5 + 4 * self._levels), dtype=np.uint32)
        if T:
            lib().orc_env_history(self._e, _p32(o))
        return o

    def get_orders(self):
        return self._book.get_orders()

    def get_trades(self):
        return self._book.get_trades()

    def get_market_data(self):
        # ref rust/src/step_sim.rs:562-607 / step_sim_numpy.rs:471-516
        h = self.history()
        d = {
            "bid_price": h[:, 1].copy(), "ask_price": h[:, 2].copy(),
            "bid_vol": h[:, 4].copy(), "ask_vol": h[:, 3].copy(), "trade_vol": h[:, 0].copy(),
        }
        for i in range(self._levels):
            d[f"bid_vol_{i}"] = h[:, 5 + 4 * i].copy()
            d[f"n_bid_{i}"] = h[:, 6 + 4 * i].copy()
            d[f"ask_vol_{i}"] = h[:, 7 + 4 * i].copy()
            d[f"n_ask_{i}"] = h[:, 8 + 4 * i].copy()
        return d


class StepEnv(_EnvBase):
    """Mirrors ``bourse.core.StepEnv`` (ref rust/src/step_sim.rs:55-607)."""

    @property
    def time(self):
        return self._book.get_time()

    @property
    def ask_vol(self):
        return int(self._l2()[3])

    @property
    def bid_vol(self):
        return int(self._l2()[4])

    @property
    def best_ask_vol(self):
        return int(self._l2()[7])

    @property
    def best_ask_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[7]), int(l2[8])

    @property
    def best_bid_vol(self):
        return int(self._l2()[5])

    @property
    def best_bid_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[5]), int(l2[6])

    @property
    def trade_vol(self):
        return self._book.trade_vol()

    @property
    def bid_ask(self):
        l2 = self._l2()
        return int(l2[1]), int(l2[2])

    def order_status(self, order_id):
        return self._book.order_status(order_id)

    def place_order(self, bid, vol, trader_id, price=None):
        return self._place(bid, vol, trader_id, price)

    def cancel_order(self, order_id):
        lib().orc_env_cancel_order(self._e, int(order_id))

    def modify_order(self, order_id, new_price=None, new_vol=None):
        lib().orc_env_modify_order(self._e, int(order_id), int(new_price is not None), int(new_price or 0),
                                   int(new_vol is not None), int(new_vol or 0))

    def get_prices(self):
        h = self.history()
        return h[:, 1].copy(), h[:, 2].copy()

    def get_volumes(self):
        h = self.history()
        return h[:, 4].copy(), h[:, 3].copy()

    def get_touch_volumes(self):
        h = self.history()
        return h[:, 5].copy(), h[:, 7].copy()

    def get_touch_order_counts(self):
        h = self.history()
        return h[:, 6].copy(), h[:, 8].copy()

    def get_trade_volumes(self):
        return self.history()[:, 0].copy()

    def level_1_data_array(self):
        # 8 values, no trade_vol (ref step_sim.rs:383-392)
        l2 = self._l2()
        return l2[1:9].copy()

    def level_2_data_array(self):
        return self._l2()


class StepEnvNumpy(_EnvBase):
    """Mirrors ``bourse.core.StepEnvNumpy`` (ref rust/src/step_sim_numpy.rs:66-516)."""

    def submit_limit_orders(self, orders):
        sides, vols, traders, prices = orders
        ids = []
        for i in range(len(sides)):  # short-circuits at the first error, earlier ones stay queued
            ids.append(self._place(bool(sides[i]), int(vols[i]), int(traders[i]), int(prices[i])))
        return np.array(ids, dtype=np.uint64)

    def submit_cancellations(self, order_ids):
        for i in np.asarray(order_ids):
            lib().orc_env_cancel_order(self._e, int(i))

    def submit_instructions_native(self, instructions):
        """``submit_instructions`` with the loop over the arrays in the library (as the reference's runs in Rust,
        rust/src/step_sim_numpy.rs:233-275) - what bench.py's INGRESS CPU baseline times.  Same ids, same stop at a bad price."""
        action, sides, vols, traders, prices, order_ids = [np.ascontiguousarray(a, dtype=t) for a, t in zip(
            instructions, (np.uint32, np.uint8, np.uint32, np.uint32, np.uint32, np.uint64))]
        n = len(action)
        ids = np.empty(n, dtype=np.uint64)
        applied = C.c_uint64(0)
        f = lib().orc_env_submit_instructions
        f.restype = C.c_int
        vp = C.c_void_p
        f.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
        rc = f(self._e, n, *[a.ctypes.data for a in (action, sides, vols, traders, prices, order_ids)], ids.ctypes.data, C.byref(applied))
        if rc == 1:
            raise _price_error(int(prices[applied.value]), self._tick)
        return ids

    def submit_instructions(self, instructions):
        action, sides, vols, traders, prices, order_ids = instructions
        ids = []
        for i in range(len(action)):
            a = int(action[i])
            if a == 1:
                ids.append(self._place(bool(sides[i]), int(vols[i]), int(traders[i]), int(prices[i])))
            elif a == 2:
                lib().orc_env_cancel_order(self._e, int(order_ids[i]))
                ids.append(U64_MAX)
            else:
                ids.append(U64_MAX)
        return np.array(ids, dtype=np.uint64)

    def level_1_data(self):
        return self._l2()[:9].copy()

    def level_2_data(self):
        return self._l2()


# ---------------------------------------------------------------------- agents
AGENT_DESC_DTYPE = np.dtype([
    ("type", "<u4"), ("n", "<u4"), ("tick_lo", "<u4"), ("tick_hi", "<u4"), ("vol_lo", "<u4"), ("vol_hi", "<u4"),
    ("tick_size", "<u4"), ("rate", "<f4"), ("trader_start", "<u4"), ("p_limit", "<f4"), ("p_market", "<f4"),
    ("p_cancel", "<f4"), ("trade_vol", "<u4"), ("pad", "<u4"), ("mu", "<f8"), ("sigma", "<f8"), ("decay", "<f8"),
    ("demand", "<f8"), ("scale", "<f8"), ("order_ratio", "<f8"),
])
assert AGENT_DESC_DTYPE.itemsize == 104


def agent_descs(members):
    """members: list of tuples
       ("random", n, (tick_lo, tick_hi), (vol_lo, vol_hi), tick_size, activity_rate)          RandomAgents::new
       ("noise", trader_start, n, dict(tick_size, p_limit, p_market, p_cancel, trade_vol, price_dist_mu, price_dist_sigma))
       ("momentum", trader_start, n, dict(tick_size, p_cancel, trade_vol, decay, demand, scale, order_ratio,
                                          price_dist_mu, price_dist_sigma))"""
    d = np.zeros(len(members), dtype=AGENT_DESC_DTYPE)
    for i, m in enumerate(members):
        if m[0] == "random":
            _, n, tr, vr, ts, rate = m
            d[i]["type"], d[i]["n"], d[i]["tick_size"], d[i]["rate"] = 0, n, ts, np.float32(rate)
            d[i]["tick_lo"], d[i]["tick_hi"], d[i]["vol_lo"], d[i]["vol_hi"] = tr[0], tr[1], vr[0], vr[1]
        else:
            kind, start, n, p = m
            d[i]["type"] = 1 if kind == "noise" else 2
            d[i]["trader_start"], d[i]["n"], d[i]["tick_size"] = start, n, p["tick_size"]
            d[i]["p_cancel"], d[i]["trade_vol"] = np.float32(p["p_cancel"]), p["trade_vol"]
            d[i]["mu"], d[i]["sigma"] = p["price_dist_mu"], p["price_dist_sigma"]
            if kind == "noise":
                d[i]["p_limit"], d[i]["p_market"] = np.float32(p["p_limit"]), np.float32(p["p_market"])
            else:
                for k in ("decay", "demand", "scale", "order_ratio"):
                    d[i][k] = p[k]
    return d


class AgentSet:
    """A derive(AgentSet) struct: members of any built-in type, updated in declaration order."""

    def __init__(self, members):
        self._a = lib().orc_agents_new()
        self.members = list(members)
        self._descs = agent_descs(self.members)
        for i in range(len(self._descs)):
            lib().orc_agents_add_desc(self._a, self._descs[i:i + 1].ctypes.data_as(C.c_void_p))

    def __del__(self):
        if getattr(self, "_a", None):
            lib().orc_agents_free(self._a)
            self._a = None

    def update(self, env):
        lib().orc_agents_update(self._a, env._e)

    def set_noise_prob(self, g, which, value):
        lib().orc_agents_set_noise_prob(self._a, g, {"p_limit": 0, "p_market": 1, "p_cancel": 2}[which],
                                        C.c_float(float(np.float32(value))))

    def order_list(self, g):
        out = np.zeros(65536, dtype=np.uint64)
        n = int(lib().orc_agents_order_list(self._a, g, _p64(out), len(out)))
        return out[:n].copy()


class RandomAgentSet:
    """An AgentSet of RandomAgents groups, updated in declaration order."""

    def __init__(self, groups):
        """groups: iterable of (n, (tick_lo, tick_hi), (vol_lo, vol_hi), tick_size, activity_rate)"""
        self._a = lib().orc_agents_new()
        self.groups = list(groups)
        for n, tr, vr, ts, rate in self.groups:
            lib().orc_agents_add_random(self._a, int(n), int(tr[0]), int(tr[1]), int(vr[0]), int(vr[1]), int(ts),
                                        C.c_float(float(np.float32(rate))))

    def __del__(self):
        if getattr(self, "_a", None):
            lib().orc_agents_free(self._a)
            self._a = None

    def held_ids(self, g):
        out = np.zeros(self.groups[g][0], dtype=np.uint64)
        lib().orc_agents_held_ids(self._a, g, _p64(out))
        return out

    def update(self, env):
        """agents.update(env, rng) with the env's own RNG."""
        lib().orc_agents_update(self._a, env._e)


def sim_runner(env, agents, seed, n_steps, rng_state=None):
    """ref crates/step_sim/src/runner.rs:46-69.  Returns the RNG state after the run."""
    st = np.zeros(2, dtype=np.uint64)
    if rng_state is None:
        lib().orc_rng_seed(int(seed), _p64(st))
    else:
        st[:] = rng_state
    rc = lib().orc_sim_run(env._e, agents._a, _p64(st), int(n_steps))
    if rc != 0:
        raise RuntimeError(f"oracle sim_run status {rc}")
    return st


class ManyBooks:
    """B independent (Env, RandomAgents groups, RNG) simulations; book b seeded seed + b."""

    def __init__(self, n_books, seed, start_time, tick_size, step_size, trading, levels, groups=None, members=None,
                 build_threads=1):
        self.n_books, self.levels, self.tick = int(n_books), int(levels), int(tick_size)
        lib().orc_many_set_build_threads(int(build_threads))
        if members is not None:  # arbitrary AgentSet (see agent_descs)
            d = agent_descs(members)
            self._m = lib().orc_many_new_mixed(self.n_books, int(seed), int(start_time), int(tick_size),
                                                int(step_size), int(bool(trading)), self.levels, len(d),
                                                d.ctypes.data_as(C.c_void_p))
            return
        g = np.zeros((len(groups), 7), dtype=np.uint32)
        for i, (n, tr, vr, ts, rate) in enumerate(groups):
            g[i, :6] = (n, tr[0], tr[1], vr[0], vr[1], ts)
            g[i, 6] = np.float32(rate).view(np.uint32)
        self._m = lib().orc_many_new(self.n_books, int(seed), int(start_time), int(tick_size), int(step_size),
                                     int(bool(trading)), self.levels, len(groups), _p32(g))

    def __del__(self):
        if getattr(self, "_m", None):
            lib().orc_many_free(self._m)
            self._m = None

    def run(self, n_steps, n_threads=1):
        rc = lib().orc_many_run(self._m, int(n_steps), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"oracle many_run status {rc}")

    def n_steps(self):
        return int(lib().orc_many_n_steps(self._m))

    def history(self, first_step=0, n=None):
        """u32[n, B, 5+4L]"""
        if n is None:
            n = self.n_steps() - first_step
        o = np.zeros((n, self.n_books, 5 + 4 * self.levels), dtype=np.uint32)
        if n:
            lib().orc_many_history(self._m, int(first_step), int(n), _p32(o))
        return o

    def book(self, b):
        return _BookView(lib().orc_many_book(self._m, int(b)), self.levels, self.tick)

    def rng_states(self):
        out = np.zeros((self.n_books, 2), dtype=np.uint64)
        for b in range(self.n_books):
            lib().orc_many_rng_state(self._m, b, _p64(out[b]))
        return out

    def trade_counts(self):
        out = np.zeros(self.n_books, dtype=np.uint64)
        lib().orc_many_trade_counts(self._m, _p64(out))
        return out

    def order_counts(self):
        out = np.zeros(self.n_books, dtype=np.uint64)
        lib().orc_many_order_counts(self._m, _p64(out))
        return out


class ManyMarkets:
    """n_markets independent (MarketEnv<assets>, RandomMarketAgents groups, RNG) simulations (ref
    crates/step_sim/src/market_env.rs, runner.rs:108-131); market m seeded seed + m.  Books are addressed flat as
    market * assets + asset.  groups: (asset, n, (tick_lo, tick_hi), (vol_lo, vol_hi), tick_size, rate)."""

    def __init__(self, n_markets, seed, start_time, tick_sizes, step_size, trading, levels, groups=(), members=None):
        self.n_markets, self.assets, self.levels = int(n_markets), len(tick_sizes), int(levels)
        self.n_books = self.n_markets * self.assets
        self.ticks = [int(t) for t in tick_sizes]
        if members is not None:  # [(asset, member)], member as in agent_descs(): Random / Noise / Momentum market twins
            d = agent_descs([m for _, m in members])
            asset = np.asarray([a for a, _ in members], dtype=np.uint32)
            tk = np.asarray(self.ticks, dtype=np.uint32)
            self._m = lib().orc_mkts_new_mixed(self.n_markets, int(seed), int(start_time), self.assets, _p32(tk),
                                               int(step_size), int(bool(trading)), self.levels, len(d),
                                               d.ctypes.data_as(C.c_void_p), _p32(asset))
            return
        g = np.zeros((max(len(groups), 1), 8), dtype=np.uint32)
        for i, (asset, n, tr, vr, ts, rate) in enumerate(groups):
            g[i, :7] = (asset, n, tr[0], tr[1], vr[0], vr[1], ts)
            g[i, 7] = np.float32(rate).view(np.uint32)
        tk = np.asarray(self.ticks, dtype=np.uint32)
        self._m = lib().orc_mkts_new(self.n_markets, int(seed), int(start_time), self.assets, _p32(tk), int(step_size),
                                     int(bool(trading)), self.levels, len(groups), _p32(g))

    def __del__(self):
        if getattr(self, "_m", None):
            lib().orc_mkts_free(self._m)
            self._m = None

    def run(self, n_steps, n_threads=1):
        rc = lib().orc_mkts_run(self._m, int(n_steps), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"oracle mkts_run status {rc}")

    def place_order(self, market, asset, bid, vol, trader_id, price=None):
        out = C.c_uint64(0)
        rc = lib().orc_mkts_place(self._m, int(market), int(asset), int(bool(bid)), int(vol), int(trader_id),
                                  int(price is not None), int(price or 0), C.byref(out))
        if rc == 1:
            raise ValueError(f"Price {price} was not a multiple of tick-size {self.ticks[asset]}")
        return int(out.value)

    def cancel_order(self, market, asset, order_id):
        lib().orc_mkts_cancel(self._m, int(market), int(asset), int(order_id))

    def modify_order(self, market, asset, order_id, new_price=None, new_vol=None):
        lib().orc_mkts_modify(self._m, int(market), int(asset), int(order_id), int(new_price is not None),
                              int(new_price or 0), int(new_vol is not None), int(new_vol or 0))

    def step(self):
        rc = lib().orc_mkts_step(self._m)
        if rc != 0:
            raise RuntimeError(f"oracle mkts_step status {rc}")

    def set_trading(self, on):
        lib().orc_mkts_set_trading(self._m, int(bool(on)))

    def n_steps(self):
        return int(lib().orc_mkts_n_steps(self._m))

    def history(self, first_step=0, n=None):
        """u32[n, n_markets * assets, 5+4L]"""
        if n is None:
            n = self.n_steps() - first_step
        o = np.zeros((n, self.n_books, 5 + 4 * self.levels), dtype=np.uint32)
        if n:
            lib().orc_mkts_history(self._m, int(first_step), int(n), _p32(o))
        return o

    def book(self, market, asset):
        return _BookView(lib().orc_mkts_book(self._m, int(market), int(asset)), self.levels, self.ticks[asset])

    def rng_states(self):
        out = np.zeros((self.n_markets, 2), dtype=np.uint64)
        for b in range(self.n_markets):
            lib().orc_mkts_rng_state(self._m, b, _p64(out[b]))
        return out


# ---------------------------------------------------------------------------------------------------------------
# Batched SoA CPU implementation (oracle/bourse_soa.cpp): ladder + per-level FIFO instead of ordered maps.  Test and
# measurement infrastructure like the rest of this directory; tests/test_soa_cpu.py checks it equal to ManyBooks.
_soa = None


def soa_lib() -> C.CDLL:
    global _soa
    if _soa is not None:
        return _soa
    build()
    L = C.CDLL(_SOA_PATH)
    u64, u32, i32, vp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p
    p64, p32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.soa_new.restype = vp
    L.soa_new.argtypes = [u32, u64, u64, u32, u64, u32, i32, p32, u64, u32, i32, i32]
    L.soa_free.restype = None
    L.soa_free.argtypes = [vp]
    L.soa_run.restype = i32
    L.soa_run.argtypes = [vp, u64, i32]
    L.soa_steps_done.restype = u64
    L.soa_steps_done.argtypes = [vp]
    L.soa_history.restype = None
    L.soa_history.argtypes = [vp, u64, u64, p32]
    for name in ("soa_trade_counts", "soa_event_counts", "soa_rng_states"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [vp, p64]
    L.soa_trades_retained.restype = u64
    L.soa_trades_retained.argtypes = [vp, u32]
    L.soa_trades.restype = None
    L.soa_trades.argtypes = [vp, u32, vp]
    L.soa_clear_trades.restype = None
    L.soa_clear_trades.argtypes = [vp]
    _soa = L
    return L


def activity_threshold(rate) -> int:
    """`gen::<f32>() < rate` as a threshold on (u32 >> 8): ceil(f32(rate) * 2^24) clamped to [0, 2^24] (the integer the
    device uses, bourse_amd/csrc/host_math.hpp; exact in double)."""
    import math

    r = float(np.float32(rate))
    if not (r > 0.0):
        return 0
    x = r * 16777216.0
    return 16777216 if x >= 16777216.0 else int(math.ceil(x))


class SoaBooks:
    """B independent RandomAgents books on the SoA CPU engine; same constructor meaning as ManyBooks (trading enabled)."""

    def __init__(self, n_books, seed, start_time, tick_size, step_size, levels, groups, history_capacity, trade_reserve=0,
                 keep_trades=True, threads=1):
        self.n_books, self.levels, self.threads = int(n_books), int(levels), int(threads)
        self.hist_cap = int(history_capacity)
        g = np.zeros((len(groups), 7), dtype=np.uint32)
        for i, (n, tr, vr, ts, rate) in enumerate(groups):
            g[i] = (n, tr[0], tr[1], vr[0], vr[1], ts, activity_threshold(rate))
        self._m = soa_lib().soa_new(self.n_books, int(seed), int(start_time), int(tick_size), int(step_size), self.levels,
                                    len(groups), _p32(g), self.hist_cap, int(trade_reserve), int(bool(keep_trades)),
                                    self.threads)
        if not self._m:
            raise ValueError("shape outside the SoA engine's ladder (tick grid / window / agent count)")

    def __del__(self):
        if getattr(self, "_m", None):
            soa_lib().soa_free(self._m)
            self._m = None

    def run(self, n_steps, n_threads=None):
        soa_lib().soa_run(self._m, int(n_steps), int(n_threads or self.threads))

    def history(self, first_step=None, n=None):
        done = int(soa_lib().soa_steps_done(self._m))
        if first_step is None:
            first_step = max(0, done - self.hist_cap)
        if n is None:
            n = done - first_step
        o = np.zeros((n, self.n_books, 5 + 4 * self.levels), dtype=np.uint32)
        if n:
            soa_lib().soa_history(self._m, int(first_step), int(n), _p32(o))
        return o

    def trade_counts(self):
        out = np.zeros(self.n_books, dtype=np.uint64)
        soa_lib().soa_trade_counts(self._m, _p64(out))
        return out

    def event_counts(self):
        out = np.zeros(self.n_books, dtype=np.uint64)
        soa_lib().soa_event_counts(self._m, _p64(out))
        return out

    def rng_states(self):
        out = np.zeros((self.n_books, 2), dtype=np.uint64)
        soa_lib().soa_rng_states(self._m, _p64(out))
        return out

    def trades(self, book):
        n = int(soa_lib().soa_trades_retained(self._m, int(book)))
        a = np.zeros(n, dtype=TRADE_DTYPE)
        if n:
            soa_lib().soa_trades(self._m, int(book), a.ctypes.data_as(C.c_void_p))
        return a

    def clear_trades(self):
        soa_lib().soa_clear_trades(self._m)
