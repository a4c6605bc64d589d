"""GPU-backed counterparts of the reference's ``bourse.core`` classes (one book per object).

* ``StepEnv``      <- rust/src/step_sim.rs:55-607
* ``StepEnvNumpy`` <- rust/src/step_sim_numpy.rs:66-516
* ``OrderBook``    <- rust/src/order_book.rs:30-380 (immediate mode; no JSON snapshot)

Same constructor, method and property names, dtypes and array layouts (the CODE's layouts:
see SURVEY §8b for the two places where the reference's docstrings disagree with its code), so
agents written against ``BaseAgent`` / ``BaseNumpyAgent`` drop in.  LEVELS is 10 as in the bindings.
Everything executes on the MI355X through the C ABI; there is no CPU path.
"""
from __future__ import annotations

import numpy as np

from .env import ManyBookEnv

LEVELS = 10


class _TradeArchive:
    """The reference keeps every trade forever (``OrderBook.trades``, orderbook.rs:107); the device retains
    ``trade_capacity`` records per book.  The one-book wrappers drain them into a host archive once half the capacity is
    in use (the strict check of every step already reads the retained count), so a long run never reaches
    BK_FLAG_TRADE_OVERFLOW unless a single step makes more than trade_capacity / 2 trades."""

    def _init_trades(self, trade_capacity):
        self._tr_cap = int(trade_capacity)
        self._tr_archive = []

    def _maybe_drain_trades(self):
        if self._tr_cap and 2 * self._env.last_retained_trades >= self._tr_cap:
            a = self._env.trades(0)
            if len(a):
                self._tr_archive.append(a)
                self._env.clear_trades()

    def _all_trades(self) -> np.ndarray:
        cur = self._env.trades(0)
        return np.concatenate(self._tr_archive + [cur]) if self._tr_archive else cur


class _EnvBase(_TradeArchive):
    def __init__(self, seed, start_time, tick_size, step_size, trading=True, *, max_live_orders=512,
                 max_orders=1 << 16, trade_capacity=1 << 16, history_capacity=1 << 12, device=0):
        self._env = ManyBookEnv(1, seed, start_time, tick_size, step_size, trading, levels=LEVELS,
                                max_live_orders=max_live_orders, max_orders=max_orders,
                                trade_capacity=trade_capacity, history_capacity=history_capacity, device=device)
        self._init_trades(trade_capacity)
        self._l2_cache = None
        # The reference keeps every step's record forever (Level2DataRecords, data.rs:26-56); the device keeps a ring
        # of history_capacity steps, drained into this host archive before it wraps.
        self._hist_cap = int(history_capacity)
        self._hist_archive = []
        self._hist_pending = 0

    # -- shared
    def enable_trading(self):
        self._env.enable_trading()

    def disable_trading(self):
        self._env.disable_trading()

    def step(self):
        self._env.step()
        self._maybe_drain_trades()
        self._l2_cache = None
        self._hist_pending += 1
        if self._hist_cap and self._hist_pending >= self._hist_cap:
            self._drain_history()

    def _drain_history(self):
        if self._hist_pending:
            self._hist_archive.append(self._env.history()[:, 0, :].copy())
            self._env.clear_history()
            self._hist_pending = 0

    def _l2(self) -> np.ndarray:
        """[trade_vol(live), bid, ask, ask_vol, bid_vol, levels...] (step_sim_numpy.rs:353-365)."""
        if self._l2_cache is None:
            self._l2_cache = self._env.level2(0, 1)[0]
        return self._l2_cache.copy()

    def get_orders(self):
        # PyOrder tuples (rust/src/types.rs:17-31)
        return [
            (bool(r["side"]), int(r["status"]), int(r["arr_time"]), int(r["end_time"]), int(r["vol"]),
             int(r["start_vol"]), int(r["price"]), int(r["trader_id"]), int(r["order_id"]))
            for r in self._env.orders(0)
        ]

    def get_trades(self):
        # PyTrade tuples (rust/src/types.rs:4-15)
        return [
            (int(r["t"]), bool(r["side"]), int(r["price"]), int(r["vol"]), int(r["active_id"]), int(r["passive_id"]))
            for r in self._all_trades()
        ]

    def _history(self) -> np.ndarray:
        if not self._hist_archive:
            return self._env.history()[:, 0, :]
        return np.concatenate(self._hist_archive + [self._env.history()[:, 0, :]], axis=0)

    def get_market_data(self):
        # key set and layout: rust/src/step_sim.rs:562-607
        h = self._history()
        d = {
            "bid_price": h[:, 1].copy(), "ask_price": h[:, 2].copy(),
            "bid_vol": h[:, 4].copy(), "ask_vol": h[:, 3].copy(), "trade_vol": h[:, 0].copy(),
        }
        for i in range(LEVELS):
            d[f"bid_vol_{i}"] = h[:, 5 + 4 * i].copy()
            d[f"n_bid_{i}"] = h[:, 6 + 4 * i].copy()
            d[f"ask_vol_{i}"] = h[:, 7 + 4 * i].copy()
            d[f"n_ask_{i}"] = h[:, 8 + 4 * i].copy()
        return d


class StepEnv(_EnvBase):
    """``bourse.core.StepEnv(seed, start_time, tick_size, step_size, trading=True)``."""

    @property
    def time(self):
        return self._env.time(0)

    @property
    def ask_vol(self):
        return int(self._l2()[3])

    @property
    def best_ask_vol(self):
        return int(self._l2()[7])

    @property
    def best_ask_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[7]), int(l2[8])

    @property
    def bid_vol(self):
        return int(self._l2()[4])

    @property
    def best_bid_vol(self):
        return int(self._l2()[5])

    @property
    def best_bid_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[5]), int(l2[6])

    @property
    def trade_vol(self):
        return self._env.trade_vol(0)

    @property
    def bid_ask(self):
        l2 = self._l2()
        return int(l2[1]), int(l2[2])

    def order_status(self, order_id):
        return self._env.order_status(0, order_id)

    def place_order(self, bid, vol, trader_id, price=None):
        return self._env.place_order(0, bid, vol, trader_id, price)

    def cancel_order(self, order_id):
        self._env.cancel_order(0, order_id)

    def modify_order(self, order_id, new_price=None, new_vol=None):
        self._env.modify_order(0, order_id, new_price, new_vol)

    def get_prices(self):
        h = self._history()
        return h[:, 1].copy(), h[:, 2].copy()

    def get_volumes(self):
        h = self._history()
        return h[:, 4].copy(), h[:, 3].copy()

    def get_touch_volumes(self):
        h = self._history()
        return h[:, 5].copy(), h[:, 7].copy()

    def get_touch_order_counts(self):
        h = self._history()
        return h[:, 6].copy(), h[:, 8].copy()

    def get_trade_volumes(self):
        return self._history()[:, 0].copy()

    def level_1_data_array(self):
        return self._l2()[1:9].copy()  # 8 values, no trade_vol (step_sim.rs:383-392)

    def level_2_data_array(self):
        return self._l2()


class OrderBook(_TradeArchive):
    """``bourse.core.OrderBook(start_time, tick_size, trading=True)`` — immediate-mode book
    (ref rust/src/order_book.rs:30-380) on the GPU: every call is one event processed at the book's current
    time (``Env::step`` with a one-event queue and step_size 0), so ``set_time`` is the caller's job exactly as in
    the reference.  Orders placed at the SAME time keep strict FIFO priority here (the reference's
    ``(price, t)`` key would overwrite).  ``save_json_snapshot`` / ``order_book_from_json`` use the
    reference's serde layout (orderbook.rs:93-112, 811-918)."""

    def __init__(self, start_time, tick_size, trading=True, *, max_live_orders=512, max_orders=1 << 16,
                 trade_capacity=1 << 16, device=0):
        self._env = ManyBookEnv(1, 0, start_time, tick_size, 0, trading, levels=LEVELS,
                                max_live_orders=max_live_orders, max_orders=max_orders,
                                trade_capacity=trade_capacity, history_capacity=0, device=device)
        self._init_trades(trade_capacity)
        self._trading = bool(trading)
        self._trade_vol0 = 0  # OrderBook.trade_vol is never reset by this class: base + volume of all trades

    def trade_vol(self):
        """Cumulative traded volume (``OrderBook::get_trade_vol``, orderbook.rs:314-316), wrapping u32."""
        return (self._trade_vol0 + int(self._all_trades()["vol"].sum(dtype=np.uint64))) & 0xFFFFFFFF

    def save_json_snapshot(self, path, pretty=False):
        """``OrderBook::save_json`` (orderbook.rs:811-819; rust/src/order_book.rs:364-380)."""
        import json

        state = self._env.book_state(0, trading=self._trading, trade_vol=self.trade_vol(), trades=self._all_trades())
        with open(path, "w") as f:
            if pretty:
                json.dump(state, f, indent=2)
            else:
                json.dump(state, f, separators=(",", ":"))

    @classmethod
    def _from_state(cls, state, **kw):
        ob = cls(int(state["t"]), int(state["tick_size"]), bool(state["trading"]), **kw)
        ob._env.load_book_state(0, state)
        vols = sum(int(t["vol"]) for t in state["trades"])
        ob._trade_vol0 = (int(state["trade_vol"]) - vols) & 0xFFFFFFFF
        return ob

    def _l2(self):
        return self._env.level2(0, 1)[0]

    def set_time(self, t):
        self._env.set_time(0, t)

    def enable_trading(self):
        self._env.enable_trading()
        self._trading = True

    def disable_trading(self):
        self._env.disable_trading()
        self._trading = False

    def ask_vol(self):
        return int(self._l2()[3])

    def best_ask_vol(self):
        return int(self._l2()[7])

    def best_ask_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[7]), int(l2[8])

    def bid_vol(self):
        return int(self._l2()[4])

    def best_bid_vol(self):
        return int(self._l2()[5])

    def best_bid_vol_and_orders(self):
        l2 = self._l2()
        return int(l2[5]), int(l2[6])

    def bid_ask(self):
        l2 = self._l2()
        return int(l2[1]), int(l2[2])

    def order_status(self, order_id):
        return self._env.order_status(0, order_id)

    def place_order(self, bid, vol, trader_id, price=None):
        """``create_and_place_order`` (ref orderbook.rs:411-421)."""
        oid = self._env.place_order(0, bid, vol, trader_id, price)
        self._env.step()
        self._maybe_drain_trades()
        return oid

    def cancel_order(self, order_id):
        self._env.cancel_order(0, order_id)
        self._env.step()

    def modify_order(self, order_id, new_price=None, new_vol=None):
        self._env.modify_order(0, order_id, new_price, new_vol)
        self._env.step()
        self._maybe_drain_trades()

    def get_trades(self):
        return [
            (int(r["t"]), bool(r["side"]), int(r["price"]), int(r["vol"]), int(r["active_id"]), int(r["passive_id"]))
            for r in self._all_trades()
        ]

    def get_orders(self):
        return [
            (bool(r["side"]), int(r["status"]), int(r["arr_time"]), int(r["end_time"]), int(r["vol"]),
             int(r["start_vol"]), int(r["price"]), int(r["trader_id"]), int(r["order_id"]))
            for r in self._env.orders(0)
        ]


def order_book_from_json(path, **kw):
    """``bourse.core.order_book_from_json(path)`` (rust/src/order_book.rs:383-398): a book on the GPU initialised from a
    snapshot written by ``save_json_snapshot`` here or by the reference's ``OrderBook::save_json``."""
    import json

    with open(path) as f:
        return OrderBook._from_state(json.load(f), **kw)


class StepEnvNumpy(_EnvBase):
    """``bourse.core.StepEnvNumpy(seed, start_time, tick_size, step_size, trading=True)``."""

    def submit_limit_orders(self, orders):
        sides, vols, traders, prices = orders
        n = len(sides)
        ins = (np.ones(n, dtype=np.uint32), sides, vols, traders, prices, np.zeros(n, dtype=np.uint64))
        return self._env.submit_instructions(0, ins)

    def submit_cancellations(self, order_ids):
        for i in np.asarray(order_ids):
            self._env.cancel_order(0, int(i))

    def submit_instructions(self, instructions):
        return self._env.submit_instructions(0, instructions)

    def level_1_data(self):
        return self._l2()[:9].copy()

    def level_2_data(self):
        return self._l2()
