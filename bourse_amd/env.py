"""``ManyBookEnv``: B independent ``bourse_de::Env`` instances stepped in lockstep on one MI355X.

Host-side mirror of the reference's Rust surface for the hot path
(``Env`` crates/step_sim/src/env.rs:58-295, ``RandomAgents`` agents/random_agent.rs:48-120,
``sim_runner`` runner.rs:46-69), calling the HIP kernels through the C ABI
(include/bourse_amd.h).  Book ``b`` owns the RNG stream
``Xoroshiro128StarStar::seed_from_u64(seed + book_offset + b)``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import AgentDesc, Config, RandomAgentsCfg, Stats, check

MAX_PRICE = 2**32 - 1


@dataclass(frozen=True)
class RandomAgents:
    """``RandomAgents::new(n_agents, tick_range, vol_range, tick_size, activity_rate)``
    (ref agents/random_agent.rs:67-81).  One instance describes one group, replicated per book."""

    n_agents: int
    tick_range: Tuple[int, int]
    vol_range: Tuple[int, int]
    tick_size: int
    activity_rate: float

    def as_tuple(self):
        return (self.n_agents, tuple(self.tick_range), tuple(self.vol_range), self.tick_size, self.activity_rate)


@dataclass(frozen=True)
class RandomMarketAgents:
    """``RandomMarketAgents::new(asset, n_agents, tick_range, vol_range, tick_size, activity_rate)``
    (ref agents/random_agent.rs:185-201): a RandomAgents group trading one asset of a market."""

    asset: int
    n_agents: int
    tick_range: Tuple[int, int]
    vol_range: Tuple[int, int]
    tick_size: int
    activity_rate: float

    def as_tuple(self):
        return (self.asset, self.n_agents, tuple(self.tick_range), tuple(self.vol_range), self.tick_size,
                self.activity_rate)


@dataclass(frozen=True)
class NoiseAgentParams:
    """ref crates/step_sim/src/agents/noise_agent.rs:24-46"""

    tick_size: int
    p_limit: float
    p_market: float
    p_cancel: float
    trade_vol: int
    price_dist_mu: float
    price_dist_sigma: float


@dataclass(frozen=True)
class NoiseAgent:
    """``NoiseAgent::new(agent_id_start, n_agents, params)`` (ref noise_agent.rs:110-123)."""

    agent_id_start: int
    n_agents: int
    params: NoiseAgentParams

    def as_tuple(self):
        return ("noise", self.agent_id_start, self.n_agents, dict(self.params.__dict__))


@dataclass(frozen=True)
class MomentumParams:
    """ref crates/step_sim/src/agents/momentum_agent.rs:24-60"""

    tick_size: int
    p_cancel: float
    trade_vol: int
    decay: float
    demand: float
    scale: float
    order_ratio: float
    price_dist_mu: float
    price_dist_sigma: float


@dataclass(frozen=True)
class MomentumAgent:
    """``MomentumAgent::new(agent_id_start, n_agents, params)`` (ref momentum_agent.rs:128-142)."""

    agent_id_start: int
    n_agents: int
    params: MomentumParams

    def as_tuple(self):
        return ("momentum", self.agent_id_start, self.n_agents, dict(self.params.__dict__))


_STICKY_MAGIC = b"BKSTICKY"  # trailer of ManyBookEnv.checkpoint(): flag bits already reported to a strict caller


class ManyBookEnv:
    """B books on one GPU.  ``Env::new(start_time, tick_size, step_size, trading)`` per book."""

    def __init__(self, n_books: int, seed: int, start_time: int, tick_size: int, step_size: int, trading: bool = True,
                 levels: int = 10, max_live_orders: int = 128, max_orders: int = 0, trade_capacity: int = 4096,
                 history_capacity: int = 0, book_offset: int = 0, device: int = 0, stream: Optional[int] = None,
                 assets: int = 1, tick_sizes: Optional[Sequence[int]] = None, strict: bool = True):
        self._L = _lib.load()
        # strict: step() and synchronous run() raise when a book NEWLY reports a capacity flag.  The reference's book is
        # unbounded (crates/order_book/src/orderbook.rs:113-115); here pool / trade-record / order-log capacities are
        # fixed, and an overflow must never pass silently.  strict=False leaves the sticky flags to flags().
        # The device flags stay sticky (flags() is the record); the exception reports each bit of each book ONCE: a
        # caller that catches it can keep stepping (clear_flags() re-arms it).  BK_FLAG_STEP_SIZE is a warning, not an
        # error: the reference's Env::step never checks the event count against step_size (env.rs:116-134).
        self.strict = bool(strict)
        self._flags_sticky = None    # per-book bits a strict check has reported and moved off the device (see raise_on_flags)
        self._warned_bits = 0        # warning-only bits (STEP_SIZE) already warned about: left on the device, not re-polled
        self.last_retained_trades = 0  # largest number of records a book retained at the last strict check
        cfg = Config()
        cfg.assets = int(assets)
        self.assets = max(1, int(assets))
        cfg.n_books, cfg.levels = int(n_books), int(levels)
        cfg.start_time, cfg.tick_size, cfg.step_size = int(start_time), int(tick_size), int(step_size)
        cfg.trading, cfg.seed, cfg.book_offset = int(bool(trading)), int(seed) & (2**64 - 1), int(book_offset)
        cfg.max_live_orders, cfg.max_orders = int(max_live_orders), int(max_orders)
        cfg.trade_capacity, cfg.history_capacity, cfg.device = int(trade_capacity), int(history_capacity), int(device)
        self.history_capacity = int(history_capacity)
        self._h = C.c_void_p()
        self.n_books, self.levels, self.tick_size = int(n_books), int(levels), int(tick_size)
        self.step_size, self.start_time = int(step_size), int(start_time)
        check(self._L.bk_env_create(C.byref(cfg), C.byref(self._h)))
        self.width = int(self._L.bk_l2_width(self._h))
        if stream is not None:
            check(self._L.bk_env_set_stream(self._h, C.c_void_p(stream)))
        if tick_sizes is not None:
            tk = np.asarray(list(tick_sizes), dtype=np.uint32)
            check(self._L.bk_set_tick_sizes(self._h, len(tk), _lib.p32(tk)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.bk_env_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ host-driven flow (Env methods)
    def place_order(self, book: int, bid: bool, vol: int, trader_id: int, price: Optional[int] = None) -> int:
        """``Env::place_order`` (env.rs:166-176); raises ValueError on a non-tick-multiple price."""
        out = C.c_uint64(0)
        check(self._L.bk_place_order(self._h, book, int(bool(bid)), int(vol), int(trader_id), int(price is not None),
                                     int(price if price is not None else 0), C.byref(out)))
        return int(out.value)

    def cancel_order(self, book: int, order_id: int):
        check(self._L.bk_cancel_order(self._h, book, int(order_id)))

    def modify_order(self, book: int, order_id: int, new_price: Optional[int] = None, new_vol: Optional[int] = None):
        check(self._L.bk_modify_order(self._h, book, int(order_id), int(new_price is not None), int(new_price or 0),
                                      int(new_vol is not None), int(new_vol or 0)))

    def submit_instructions(self, book: int, instructions) -> np.ndarray:
        """``StepEnvNumpy.submit_instructions`` (rust/src/step_sim_numpy.rs:233-275)."""
        action, sides, vols, traders, prices, order_ids = instructions
        action = np.ascontiguousarray(action, dtype=np.uint32)
        sides = np.ascontiguousarray(np.asarray(sides).astype(np.uint8))
        vols = np.ascontiguousarray(vols, dtype=np.uint32)
        traders = np.ascontiguousarray(traders, dtype=np.uint32)
        prices = np.ascontiguousarray(prices, dtype=np.uint32)
        order_ids = np.ascontiguousarray(order_ids, dtype=np.uint64)
        n = len(action)
        out = np.full(n, 2**64 - 1, dtype=np.uint64)
        done = C.c_size_t(0)
        check(self._L.bk_submit_instructions(self._h, book, n, _lib.p32(action), _lib.p8(sides), _lib.p32(vols),
                                             _lib.p32(traders), _lib.p32(prices), _lib.p64(order_ids),
                                             _lib.p64(out), C.byref(done)))
        return out

    def _host_batch(self, book_offsets, instructions):
        action, sides, vols, traders, prices, order_ids = instructions
        off = np.ascontiguousarray(book_offsets, dtype=np.uint64)
        if len(off) != self.n_books + 1:
            raise ValueError("book_offsets needs n_books + 1 entries")
        action = np.ascontiguousarray(action, dtype=np.uint32)
        sides = np.asarray(sides)
        sides = np.ascontiguousarray(sides if sides.dtype == np.uint8 else sides.astype(np.uint8))
        vols = np.ascontiguousarray(vols, dtype=np.uint32)
        traders = np.ascontiguousarray(traders, dtype=np.uint32)
        prices = np.ascontiguousarray(prices, dtype=np.uint32)
        order_ids = np.ascontiguousarray(order_ids, dtype=np.uint64)
        if not (len(action) == len(sides) == len(vols) == len(traders) == len(prices) == len(order_ids) == int(off[-1])):
            raise ValueError("instruction arrays must all have book_offsets[-1] elements")
        return off, action, sides, vols, traders, prices, order_ids

    def submit_instructions_all(self, book_offsets, instructions) -> np.ndarray:
        """``submit_instructions`` (rust/src/step_sim_numpy.rs:233-275) for every book in one call: book b's instructions
        are elements ``[book_offsets[b], book_offsets[b + 1])`` of the six HOST arrays; returns the ids (``2**64 - 1``
        for elements that created nothing).  On a device-ingress env the arrays travel pinned staging -> async upload ->
        ``k_ingest`` (``bk_submit_instructions_host``), and a book whose batch stopped at a bad price raises the
        reference's ``ValueError`` AFTER every book has been applied (books are independent; earlier elements of the failing
        book stay queued, as in the reference); otherwise the host half of ``Env`` walks them
        (``bk_submit_instructions_csr``: stops at the first bad price of any book)."""
        if getattr(self, "_device_ingress", False):
            out, status, bad = self.submit_result(self.submit_instructions_all_async(book_offsets, instructions))
            if bad is not None:
                code, applied = int(status[bad, 0]), int(status[bad, 1])
                if code == _lib.BK_PRICE:
                    raise ValueError(f"book {bad}: a price of its batch was not a multiple of the tick size "
                                     f"(element {applied} of the book's batch; earlier elements are queued)")
                raise _lib.CapacityError(code, f"book {bad}: event queue / id space exhausted after {applied} elements")
            return out
        off, action, sides, vols, traders, prices, order_ids = self._host_batch(book_offsets, instructions)
        out = np.full(len(action), 2**64 - 1, dtype=np.uint64)
        done = C.c_size_t(0)
        check(self._L.bk_submit_instructions_csr(self._h, _lib.p64(off), _lib.p32(action), _lib.p8(sides), _lib.p32(vols),
                                                 _lib.p32(traders), _lib.p32(prices), _lib.p64(order_ids), _lib.p64(out),
                                                 C.byref(done)))
        return out

    def submit_instructions_all_async(self, book_offsets, instructions) -> int:
        """Device-ingress env: queue the host arrays (``bk_submit_instructions_host``) and return a TICKET at once; the ids
        and the per-book status are fetched later with ``submit_result(ticket)`` (two tickets may be in flight: the upload
        of one runs under the step kernel of the other).  Arrays obtained from ``ingress_staging()`` are uploaded in
        place, anything else is first copied to pinned memory by the library's host threads."""
        off, action, sides, vols, traders, prices, order_ids = self._host_batch(book_offsets, instructions)
        t = C.c_uint64(0)
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        check(self._L.bk_submit_instructions_host(self._h, vp(off), vp(action), vp(sides), vp(vols), vp(traders), vp(prices),
                                                  vp(order_ids), C.byref(t)))
        self._ticket_n = getattr(self, "_ticket_n", {})
        self._ticket_n[int(t.value)] = len(action)
        self._ticket_n.pop(int(t.value) - 2, None)
        return int(t.value)

    def submit_result(self, ticket: int, ids: bool = True, out: Optional[np.ndarray] = None, status: Optional[np.ndarray] = None,
                      view: bool = False):
        """(ids u64[n] or None, status u32[n_books, 2] = {code, elements applied}, lowest failing book or None) of a ticket.
        ``out`` / ``status``: arrays to fill instead of fresh ones (a loop that fetches every step saves their page faults).
        ``view=True``: no copy at all - read-only numpy views of the library's pinned staging, valid until two more submits."""
        n = getattr(self, "_ticket_n", {}).get(int(ticket))
        if n is None:  # not a ticket of this env, or two submits old: the library says which
            check(self._L.bk_submit_result(self._h, int(ticket), None, None, None))
            raise _lib.BourseError(_lib.BK_INVALID, f"ticket {ticket}: its element count is no longer known")
        if view:
            pi, ps, bad = C.c_void_p(0), C.c_void_p(0), C.c_uint32(0)
            check(self._L.bk_submit_result_view(self._h, int(ticket), C.byref(pi), C.byref(ps), C.byref(bad)))

            def ro(ptr, dtype, count, shape):
                if count == 0:
                    return np.zeros(shape, dtype=dtype)
                a = np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype, count=count).reshape(shape)
                a.flags.writeable = False
                return a

            return ((ro(pi.value, np.uint64, n, (n,)) if ids else None), ro(ps.value, np.uint32, 2 * self.n_books, (self.n_books, 2)),
                    (None if bad.value == 0xFFFFFFFF else int(bad.value)))
        if ids:
            if out is None:
                out = np.empty(n, dtype=np.uint64)
            elif out.dtype != np.uint64 or not out.flags.c_contiguous or len(out) < n:
                raise ValueError("out: a contiguous uint64 array of at least the ticket's element count")
        if status is None:
            status = np.empty((self.n_books, 2), dtype=np.uint32)
        elif status.dtype != np.uint32 or not status.flags.c_contiguous or status.size != 2 * self.n_books:
            raise ValueError("status: a contiguous uint32 array of 2 x n_books")
        bad = C.c_uint32(0)
        check(self._L.bk_submit_result(self._h, int(ticket), C.c_void_p(out.ctypes.data) if ids and n else None,
                                       C.c_void_p(status.ctypes.data), C.byref(bad)))
        return (out[:n] if ids else None), status, (None if bad.value == 0xFFFFFFFF else int(bad.value))

    def ingress_staging(self, min_elements: int) -> dict:
        """Pinned staging arrays of the NEXT ``submit_instructions_all(_async)`` call (``bk_ingress_staging``), as numpy
        views: ``book_offsets`` u64[n_books + 1], ``action`` / ``vol`` / ``trader_id`` / ``price`` u32, ``side`` u8,
        ``order_id`` u64, each of ``capacity >= min_elements`` elements.  Fill them in place and pass slices of THESE arrays
        to the submit call: they are uploaded without a host copy.  Valid until that submit returns."""
        a = _lib.IngressArrays()
        check(self._L.bk_ingress_staging(self._h, int(min_elements), C.byref(a)))
        cap = int(a.capacity)

        def view(ptr, dtype, n):
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n)

        return {"capacity": cap, "book_offsets": view(a.book_offsets, np.uint64, self.n_books + 1),
                "action": view(a.action, np.uint32, cap), "side": view(a.side, np.uint8, cap),
                "vol": view(a.vol, np.uint32, cap), "trader_id": view(a.trader_id, np.uint32, cap),
                "price": view(a.price, np.uint32, cap), "order_id": view(a.order_id, np.uint64, cap)}

    def enable_trading(self):
        check(self._L.bk_enable_trading(self._h, 1))

    def disable_trading(self):
        check(self._L.bk_enable_trading(self._h, 0))

    def step(self, sync: bool = True):
        """``Env::step`` (env.rs:116-135) for every book over the queued events.  ``sync=False`` (device-resident
        ingress only): queue the step on the env's stream and return (no flag check)."""
        if not sync:
            check(self._L.bk_step_async(self._h))
            return
        check(self._L.bk_step(self._h))
        if self.strict:
            self.raise_on_flags()

    # ------------------------------------------------------------ device-resident instruction ingress
    def enable_device_ingress(self, queue_capacity: int = 256):
        """Switch this (fresh) env to instructions submitted FROM DEVICE MEMORY (``bk_device_ingress_enable``): at most
        ``queue_capacity`` events per book (market) and step."""
        check(self._L.bk_device_ingress_enable(self._h, int(queue_capacity)))
        self._device_ingress = True

    @staticmethod
    def _dev_ptr(x, itemsize, name):
        """Device pointer of a torch CUDA tensor / anything with ``__cuda_array_interface__`` (contiguous, right width)."""
        if x is None:
            return None
        if hasattr(x, "data_ptr"):  # torch
            if not x.is_cuda or not x.is_contiguous() or x.element_size() != itemsize:
                raise ValueError(f"{name}: a contiguous CUDA tensor of {itemsize}-byte elements is needed")
            return C.c_void_p(x.data_ptr())
        cai = getattr(x, "__cuda_array_interface__", None)
        if cai is None or cai.get("strides") or np.dtype(cai["typestr"]).itemsize != itemsize:
            raise ValueError(f"{name}: a contiguous device array of {itemsize}-byte elements is needed")
        return C.c_void_p(cai["data"][0])

    def submit_instructions_device(self, book_offsets, action, side, vol, trader_id, price, order_id, out_ids=None,
                                   status=None, check_status: bool = False):
        """``submit_instructions`` (rust/src/step_sim_numpy.rs:233-275) for every book from DEVICE arrays, nothing passing
        through the host: book ``b``'s instructions are elements ``[book_offsets[b], book_offsets[b+1])``.  Arguments are
        torch CUDA tensors (or any ``__cuda_array_interface__`` object): ``book_offsets`` 8-byte ints (n_books + 1),
        ``action`` / ``vol`` / ``trader_id`` / ``price`` 4-byte ints, ``side`` 1-byte (bool / uint8), ``order_id`` 8-byte.
        ``out_ids`` (8-byte, one per element) receives the created ids (2**64 - 1 elsewhere), ``status`` (4-byte,
        2 per book) ``{code, elements applied}``.  Queued on the env's stream - which must be the stream the arrays were
        written on (``ManyBookEnv(stream=torch.cuda.current_stream().cuda_stream)``).  ``check_status=True`` waits and
        raises the reference's ``ValueError`` for the first book whose batch stopped at a bad price."""
        P = self._dev_ptr
        off_ptr = P(book_offsets, 8, "book_offsets")  # (validates the first argument before anything is looked at)
        if action is None:
            raise ValueError("action: a contiguous CUDA tensor of 4-byte elements is needed")
        n_elem = action.numel() if hasattr(action, "numel") else int(np.prod(action.__cuda_array_interface__["shape"]))
        if n_elem == 0:  # nothing to submit for any book (an empty device array has no address to hand over)
            if status is not None and hasattr(status, "zero_"):
                status.zero_()
            return out_ids, status
        check(self._L.bk_submit_instructions_device(self._h, off_ptr, P(action, 4, "action"),
                                                    P(side, 1, "side"), P(vol, 4, "vol"), P(trader_id, 4, "trader_id"),
                                                    P(price, 4, "price"), P(order_id, 8, "order_id"), P(out_ids, 8, "out_ids"),
                                                    P(status, 4, "status")))
        if check_status:
            if status is None:
                raise ValueError("check_status needs a status array")
            self.sync()
            st = status.cpu().numpy() if hasattr(status, "cpu") else np.asarray(status)
            st = st.reshape(-1, 2).view(np.uint32) if st.dtype.itemsize == 4 else st.reshape(-1, 2)
            bad = np.nonzero(st[:, 0])[0]
            if len(bad):
                b, code = int(bad[0]), int(st[bad[0], 0])
                if code == _lib.BK_PRICE:
                    raise ValueError(f"book {b}: a price of its batch was not a multiple of the tick size "
                                     f"(element {int(st[b, 1])} of the book's batch; earlier elements are queued)")
                raise _lib.CapacityError(code, f"book {b}: event queue / id space exhausted after {int(st[b, 1])} elements")
        return out_ids, status

    def flags_summary(self) -> Tuple[int, int]:
        """(OR of every book's sticky flags, largest number of trade records a book retains): one small reduction on
        the device and an 8-byte copy - what the strict checks poll instead of the per-book array."""
        f, r = C.c_uint32(0), C.c_uint64(0)
        check(self._L.bk_flags_summary(self._h, C.byref(f), C.byref(r)))
        return int(f.value), int(r.value)

    def clear_flags(self, mask: int = 0xFFFFFFFF):
        """Clear sticky flag bits (on the device and in the host-side record of the bits already reported)."""
        check(self._L.bk_clear_flags(self._h, int(mask) & 0xFFFFFFFF))
        if self._flags_sticky is not None:
            self._flags_sticky &= np.uint32(~int(mask) & 0xFFFFFFFF)
        self._warned_bits &= ~int(mask) & 0xFFFFFFFF

    def raise_on_flags(self, mask: Optional[int] = None):
        """Raise ``CapacityError`` (capacity bits) / ``BourseError`` (price-tick) for flag bits in ``mask`` that a book
        carries (default mask: every flag except UNKNOWN_ORDER, which ``step`` reports itself).  A STEP_SIZE bit alone
        only warns: the reference tolerates such a step.  Reported bits MOVE from the device to a host-side sticky record
        (``flags()`` returns both): the two-word summary reads 0 again, so the next check stays cheap, and a book that
        overflows AGAIN after the caller caught the error is reported again instead of passing silently."""
        import warnings

        m = (~_lib.FLAG_UNKNOWN_ORDER & 0xFFFFFFFF) if mask is None else (int(mask) & 0xFFFFFFFF)
        # The warning-only STEP_SIZE bit STAYS on the device once it has been warned about: a workload that regularly queues
        # >= step_size events (which the reference tolerates) would otherwise re-set it every step, and every strict step
        # would fetch the per-book array, launch the clear kernel and warn again.  clear_flags() re-arms the warning.
        m &= ~self._warned_bits & 0xFFFFFFFF
        any_or, self.last_retained_trades = self.flags_summary()
        if not (any_or & m):
            return  # nothing set anywhere (the common case): the per-book array is not fetched
        f = np.zeros(self.n_books, dtype=np.uint32)
        check(self._L.bk_book_flags(self._h, _lib.p32(f)))
        new = f & np.uint32(m)
        if not new.any():
            return
        bits = int(np.bitwise_or.reduce(new))
        self._warned_bits |= bits & _lib.FLAG_STEP_SIZE
        moved = bits & ~_lib.FLAG_STEP_SIZE & 0xFFFFFFFF  # error bits move to the host-side record; the warning bit does not
        if moved:
            if self._flags_sticky is None:
                self._flags_sticky = np.zeros_like(f)
            self._flags_sticky |= new & np.uint32(moved)
            check(self._L.bk_clear_flags(self._h, moved))
        books = np.nonzero(new)[0]
        names = "; ".join(n for b, n in _lib.FLAG_NAMES.items() if bits & b)
        msg = f"{len(books)} book(s) flagged (first: book {int(books[0])}): {names}"
        if bits & _lib.CAPACITY_FLAGS:
            raise _lib.CapacityError(_lib.BK_CAPACITY, msg)
        if bits & ~_lib.FLAG_STEP_SIZE:
            raise _lib.BourseError(_lib.BK_INVALID, msg)
        warnings.warn(f"bourse_amd: {msg} (price-time keys of that step may collide; the reference does not check this)",
                      RuntimeWarning, stacklevel=3)

    def order_status(self, book: int, order_id: int) -> int:
        out = C.c_uint8(0)
        check(self._L.bk_order_status(self._h, book, int(order_id), C.byref(out)))
        return int(out.value)

    def order_count(self, book: int) -> int:
        out = C.c_uint64(0)
        check(self._L.bk_order_count(self._h, book, C.byref(out)))
        return int(out.value)

    def orders(self, book: int) -> np.ndarray:
        n = self.order_count(book)
        a = np.zeros(n, dtype=_lib.ORDER_DTYPE)
        if n:
            check(self._L.bk_get_orders(self._h, book, 0, n, a.ctypes.data_as(C.c_void_p)))
        return a

    def order_keys(self, book: int) -> Tuple[np.ndarray, np.ndarray]:
        """``OrderEntry.key`` of every order (orderbook.rs:34-39) as (key price, key time)."""
        n = self.order_count(book)
        kp, kt = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint64)
        if n:
            check(self._L.bk_get_order_keys(self._h, book, 0, n, _lib.p32(kp), _lib.p64(kt)))
        return kp, kt

    # ------------------------------------------------------------ JSON snapshots (serde layout of the reference)
    def book_state(self, book: int, trading: bool = True, trade_vol: Optional[int] = None, trades=None) -> dict:
        """The book as ``serde_json`` serialises ``OrderBook`` (orderbook.rs:93-112: t, tick_size, trade_vol, orders
        [{order, key}], trades, trading; ask_side / bid_side are skipped and rebuilt on load, :891-918)."""
        side = {1: "Bid", 0: "Ask"}
        status = ["New", "Active", "Filled", "Cancelled", "Rejected"]  # types.rs:51-63
        o, (kp, kt) = self.orders(book), self.order_keys(book)
        orders = []
        for r, p, t in zip(o, kp.tolist(), kt.tolist()):
            bid = int(r["side"])
            orders.append({
                "order": {"side": side[bid], "status": status[int(r["status"])], "arr_time": int(r["arr_time"]),
                          "end_time": int(r["end_time"]), "vol": int(r["vol"]), "start_vol": int(r["start_vol"]),
                          "price": int(r["price"]), "trader_id": int(r["trader_id"]), "order_id": int(r["order_id"])},
                "key": [side[bid], (MAX_PRICE - p) if bid else p, t],  # price_key, side.rs:300-313
            })
        trades = [{"t": int(r["t"]), "side": side[int(r["side"])], "price": int(r["price"]), "vol": int(r["vol"]),
                   "active_order_id": int(r["active_id"]), "passive_order_id": int(r["passive_id"])}
                  for r in (self.trades(book, first=0) if trades is None else trades)]
        tick = self.tick_sizes[book % self.assets] if hasattr(self, "tick_sizes") else self.tick_size
        return {"t": self.time(book), "tick_size": int(tick),
                "trade_vol": self.trade_vol(book) if trade_vol is None else int(trade_vol), "orders": orders,
                "trades": trades, "trading": bool(trading)}

    def load_book_state(self, book: int, state: dict):
        """``TryFrom<OrderBookState>`` (orderbook.rs:891-918) for one book of this env (same tick size)."""
        side = {"Bid": 1, "Ask": 0}
        status = {"New": 0, "Active": 1, "Filled": 2, "Cancelled": 3, "Rejected": 4}
        tick = self.tick_sizes[book % self.assets] if hasattr(self, "tick_sizes") else self.tick_size
        if int(state["tick_size"]) != tick:
            raise ValueError("snapshot tick_size differs from the book's")
        n = len(state["orders"])
        o = np.zeros(n, dtype=_lib.ORDER_DTYPE)
        kp, kt = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint64)
        for i, e in enumerate(state["orders"]):
            r, k = e["order"], e["key"]
            bid = side[r["side"]]
            o[i] = (bid, status[r["status"]], r["arr_time"], r["end_time"], r["vol"], r["start_vol"], r["price"],
                    r["trader_id"], r["order_id"])
            kp[i] = (MAX_PRICE - k[1]) if side[k[0]] else k[1]
            kt[i] = k[2]
        t = np.zeros(len(state["trades"]), dtype=_lib.TRADE_DTYPE)
        for i, r in enumerate(state["trades"]):
            t[i] = (r["t"], side[r["side"]], r["price"], r["vol"], r["active_order_id"], r["passive_order_id"])
        check(self._L.bk_load_book(self._h, book, int(state["t"]), int(state["trade_vol"]), n,
                                   o.ctypes.data_as(C.c_void_p), _lib.p32(kp), _lib.p64(kt), len(t),
                                   t.ctypes.data_as(C.c_void_p)))

    # ------------------------------------------------------------ on-device flow
    def set_random_agents(self, groups: Iterable[RandomAgents | tuple]):
        gs = [g.as_tuple() if isinstance(g, RandomAgents) else g for g in groups]
        arr = (RandomAgentsCfg * max(len(gs), 1))()
        for i, (n, tr, vr, ts, rate) in enumerate(gs):
            arr[i].n_agents, arr[i].tick_lo, arr[i].tick_hi = int(n), int(tr[0]), int(tr[1])
            arr[i].vol_lo, arr[i].vol_hi, arr[i].tick_size = int(vr[0]), int(vr[1]), int(ts)
            arr[i].activity_rate = float(np.float32(rate))
        check(self._L.bk_set_random_agents(self._h, len(gs), arr))
        self.groups = gs

    def set_random_market_agents(self, groups: Iterable["RandomMarketAgents | tuple"]):
        """A ``MarketAgentSet`` of ``RandomMarketAgents`` groups ``(asset, n, tick_range, vol_range, tick_size, rate)``,
        identical for every market (needs ``assets > 1`` books per market)."""
        gs = [g.as_tuple() if isinstance(g, RandomMarketAgents) else g for g in groups]
        arr = (RandomAgentsCfg * max(len(gs), 1))()
        assets = np.zeros(max(len(gs), 1), dtype=np.uint32)
        for i, (asset, n, tr, vr, ts, rate) in enumerate(gs):
            assets[i] = int(asset)
            arr[i].n_agents, arr[i].tick_lo, arr[i].tick_hi = int(n), int(tr[0]), int(tr[1])
            arr[i].vol_lo, arr[i].vol_hi, arr[i].tick_size = int(vr[0]), int(vr[1]), int(ts)
            arr[i].activity_rate = float(np.float32(rate))
        check(self._L.bk_set_random_market_agents(self._h, len(gs), arr, _lib.p32(assets)))
        self.groups = gs

    def set_agents(self, members, assets=None):
        """A ``#[derive(AgentSet)]`` struct: members updated in declaration order (ref crates/macros/src/lib.rs:57-73).
        Each member is a RandomAgents / NoiseAgent / MomentumAgent instance or the equivalent tuple
        ``("random", n, tick_range, vol_range, tick_size, rate)`` / ``("noise"|"momentum", id_start, n, params_dict)``."""
        ms = []
        for m in members:
            if isinstance(m, RandomAgents):
                ms.append(("random",) + m.as_tuple())
            elif isinstance(m, (NoiseAgent, MomentumAgent)):
                ms.append(m.as_tuple())
            else:
                ms.append(tuple(m))
        arr = (AgentDesc * max(len(ms), 1))()
        for i, m in enumerate(ms):
            d = arr[i]
            if m[0] == "random":
                _, n, tr, vr, ts, rate = m
                d.type, d.n_agents, d.tick_size, d.activity_rate = 0, int(n), int(ts), float(np.float32(rate))
                d.tick_lo, d.tick_hi, d.vol_lo, d.vol_hi = int(tr[0]), int(tr[1]), int(vr[0]), int(vr[1])
            else:
                kind, start, n, p = m
                d.type = 1 if kind == "noise" else 2
                d.agent_id_start, d.n_agents, d.tick_size = int(start), int(n), int(p["tick_size"])
                d.p_cancel, d.trade_vol = float(np.float32(p["p_cancel"])), int(p["trade_vol"])
                d.price_dist_mu, d.price_dist_sigma = float(p["price_dist_mu"]), float(p["price_dist_sigma"])
                if kind == "noise":
                    d.p_limit, d.p_market = float(np.float32(p["p_limit"])), float(np.float32(p["p_market"]))
                else:
                    d.decay, d.demand = float(p["decay"]), float(p["demand"])
                    d.scale, d.order_ratio = float(p["scale"]), float(p["order_ratio"])
        if assets is not None:
            as_arr = np.asarray(list(assets), dtype=np.uint32)
            check(self._L.bk_set_market_agents(self._h, len(ms), arr, _lib.p32(as_arr)))
        else:
            check(self._L.bk_set_agents(self._h, len(ms), arr))
        self.members = ms

    def set_market_agents(self, members):
        """A ``#[derive(MarketAgentSet)]`` struct: ``[(asset, member), ...]`` with members as in ``set_agents`` — the
        multi-asset twins RandomMarketAgents / NoiseMarketAgent / MomentumMarketAgent (needs ``assets > 1``)."""
        members = list(members)
        self.set_agents([m for _, m in members], assets=[int(a) for a, _ in members])

    def run(self, n_steps: int, sync: bool = True):
        """``sim_runner``'s loop body ``n_steps`` times in ONE kernel launch (runner.rs:53-68)."""
        check(self._L.bk_run(self._h, int(n_steps)))
        if sync:
            self.sync()
            if self.strict:
                self.raise_on_flags()

    def warm(self, n_steps: int = 100):
        """``bk_warm``: ``n_steps`` of this env's own kernels on its own books, then everything is put back (state,
        level-2 records, step counter; no history slot or trade record is written).  Brings the GPU's clocks up and
        pays the pipeline's one-off set-up before a short, latency-sensitive ``run``."""
        check(self._L.bk_warm(self._h, int(n_steps)))

    def sync(self):
        check(self._L.bk_env_sync(self._h))

    # ------------------------------------------------------------ readers
    def level2(self, first_book: int = 0, n_books: Optional[int] = None) -> np.ndarray:
        """u32[n_books, 5+4L]: end-of-step snapshot, numpy ``level_2_data`` layout."""
        n = self.n_books - first_book if n_books is None else n_books
        out = np.zeros((n, self.width), dtype=np.uint32)
        check(self._L.bk_level2(self._h, first_book, n, _lib.p32(out)))
        return out

    def history_len(self) -> Tuple[int, int]:
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.bk_history_len(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def history(self, first_step: Optional[int] = None, n_steps: Optional[int] = None, first_book: int = 0,
                n_books: Optional[int] = None) -> np.ndarray:
        """u32[n_steps, n_books, 5+4L] (Level2DataRecords + trade_vols, data.rs:9-57)."""
        f, n = self.history_len()
        first_step = f if first_step is None else first_step
        n_steps = (f + n - first_step) if n_steps is None else n_steps
        nb = self.n_books - first_book if n_books is None else n_books
        out = np.zeros((n_steps, nb, self.width), dtype=np.uint32)
        if n_steps and nb:
            check(self._L.bk_history(self._h, first_step, n_steps, first_book, nb, _lib.p32(out)))
        return out

    def stream_history(self, n_steps: int, chunk: int, on_chunk=None, trades: bool = False,
                       trade_records_per_chunk: Optional[int] = None) -> dict:
        """Run ``n_steps`` in chunks while the previous chunk's L2 records stream to pinned host memory on a copy
        stream (double-buffered; needs history_capacity >= 2 * chunk).  ``on_chunk(first_step, l2[chunk, B, W])`` is
        called once a chunk has landed (the arrays are reused two chunks later).  With ``trades=True`` the chunk's trade
        records travel too, compacted on the device into one dense stream (``bk_trades_compact``):
        ``on_chunk(first_step, l2, (offsets[B+1], records))``; needs trade_capacity >= the trades of one chunk per
        book.  Returns timing figures."""
        import time

        if chunk < 1 or self.history_capacity < 2 * chunk:
            raise ValueError("stream_history needs history_capacity >= 2 * chunk")
        L, W, B = self._L, self.width, self.n_books
        out_t = C.POINTER(C.c_uint32)
        tcap = int(trade_records_per_chunk or 64 * chunk * B) if trades else 0
        streams, bufs, tbufs = [], [], []
        for _ in range(2):  # chunk k uses stream/buffer k % 2: copy k overlaps run k+1, and is awaited before run k+2
            cs, p = C.c_void_p(), C.c_void_p()
            check(L.bk_stream_create(C.byref(cs)))
            check(L.bk_pinned_alloc(chunk * B * W * 4, C.byref(p)))
            streams.append(cs)
            bufs.append((p, np.ctypeslib.as_array(C.cast(p, out_t), shape=(chunk, B, W))))
            if trades:
                pr, po = C.c_void_p(), C.c_void_p()
                check(L.bk_pinned_alloc(tcap * 40, C.byref(pr)))
                check(L.bk_pinned_alloc((B + 1) * 8, C.byref(po)))
                rec = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint8)), shape=(tcap * 40,)).view(_lib.TRADE_DTYPE)
                off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(B + 1,))
                tbufs.append((pr, po, rec, off))
        inflight = [None, None]  # per buffer: (first_step, n_steps, n_trade_records)

        def land(i):
            if inflight[i] is not None:
                check(L.bk_stream_sync(streams[i]))
                if on_chunk is not None:
                    first, c, nt = inflight[i]
                    if trades:
                        on_chunk(first, bufs[i][1][:c], (tbufs[i][3], tbufs[i][2][:nt]))
                    else:
                        on_chunk(first, bufs[i][1][:c])
                inflight[i] = None

        done, k, nbytes = 0, 0, 0
        t0 = time.perf_counter()
        try:
            while done < n_steps:
                c = min(chunk, n_steps - done)
                i = k % 2
                land(i)  # chunk k-2 has left the ring and its host buffer has been consumed
                first = self.steps_done()
                self.run(c, sync=False)
                check(L.bk_history_copy_async(self._h, first, c, 0, B, C.cast(bufs[i][0], out_t), streams[i]))
                nt = 0
                if trades:
                    check(L.bk_stream_sync(streams[1 - i]))  # the device's dense buffer is still being copied out
                    total = C.c_uint64(0)
                    check(L.bk_trades_compact(self._h, C.byref(total)))
                    nt = int(total.value)
                    if nt > tcap:
                        raise _lib.CapacityError(f"{nt} trade records in one chunk exceed trade_records_per_chunk={tcap}")
                    check(L.bk_trades_compact_copy_async(self._h, tbufs[i][0], C.cast(tbufs[i][1], C.POINTER(C.c_uint64)),
                                                         streams[i]))
                    nbytes += nt * 40 + (B + 1) * 8
                inflight[i] = (first, c, nt)
                nbytes += c * B * W * 4
                done += c
                k += 1
            land(k % 2)
            land((k + 1) % 2)
            self.sync()
            dt = time.perf_counter() - t0
        finally:
            for cs, (p, _) in zip(streams, bufs):
                L.bk_stream_sync(cs)
                L.bk_pinned_free(p)
                L.bk_stream_destroy(cs)
            for pr, po, _, _ in tbufs:
                L.bk_pinned_free(pr)
                L.bk_pinned_free(po)
        return {"seconds": dt, "book_steps_per_s": B * n_steps / dt, "d2h_gb_per_s": nbytes / dt / 1e9, "bytes": nbytes}

    def clear_history(self):
        check(self._L.bk_clear_history(self._h))

    def trade_count(self, book: int) -> Tuple[int, int]:
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.bk_trade_count(self._h, book, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def trade_counts(self) -> np.ndarray:
        out = np.zeros(self.n_books, dtype=np.uint64)
        check(self._L.bk_trade_counts(self._h, _lib.p64(out)))
        return out

    def trades(self, book: int, first: Optional[int] = None, n: Optional[int] = None) -> np.ndarray:
        total, base = self.trade_count(book)
        first = base if first is None else first
        n = total - first if n is None else n
        a = np.zeros(n, dtype=_lib.TRADE_DTYPE)
        if n:
            check(self._L.bk_get_trades(self._h, book, first, n, a.ctypes.data_as(C.c_void_p)))
        return a

    def drain_trades(self) -> Tuple[np.ndarray, np.ndarray]:
        """All retained trade records of all books as ONE dense array + CSR offsets (book b: ``rec[off[b]:off[b+1]]``),
        compacted on the device; the records are then consumed (like ``clear_trades``)."""
        total = C.c_uint64(0)
        check(self._L.bk_trades_compact(self._h, C.byref(total)))
        rec = np.zeros(int(total.value), dtype=_lib.TRADE_DTYPE)
        off = np.zeros(self.n_books + 1, dtype=np.uint64)
        check(self._L.bk_trades_compact_copy_async(self._h, rec.ctypes.data_as(C.c_void_p), _lib.p64(off), None))
        return off, rec

    def clear_trades(self):
        check(self._L.bk_clear_trades(self._h))

    def order_counts(self) -> np.ndarray:
        """Orders created so far per book by the on-device agents (``orders.len()``, orderbook.rs:327-329)."""
        out = np.zeros(self.n_books, dtype=np.uint64)
        check(self._L.bk_order_counts(self._h, _lib.p64(out)))
        return out

    def event_steps_keyed(self) -> np.ndarray:
        """Per book: how many host-driven / ingress steps ran on the keyed event loop (a diagnostic; ``bk_event_steps_keyed``)."""
        out = np.zeros(self.n_books, dtype=np.uint64)
        check(self._L.bk_event_steps_keyed(self._h, _lib.p64(out)))
        return out

    def time(self, book: int = 0) -> int:
        out = C.c_uint64(0)
        check(self._L.bk_time(self._h, book, C.byref(out)))
        return int(out.value)

    def set_time(self, book: int, t: int):
        """``OrderBook::set_time`` (orderbook.rs:183-185) — host-driven path."""
        check(self._L.bk_set_time(self._h, book, int(t)))

    def trade_vol(self, book: int = 0) -> int:
        out = C.c_uint32(0)
        check(self._L.bk_trade_vol(self._h, book, C.byref(out)))
        return int(out.value)

    def steps_done(self) -> int:
        out = C.c_uint64(0)
        check(self._L.bk_steps_done(self._h, C.byref(out)))
        return int(out.value)

    def flags(self) -> np.ndarray:
        """Every book's sticky flag word: the device's bits OR the ones a strict check has already reported."""
        out = np.zeros(self.n_books, dtype=np.uint32)
        check(self._L.bk_book_flags(self._h, _lib.p32(out)))
        if self._flags_sticky is not None:
            out |= self._flags_sticky
        return out

    def rng_state(self, book: int) -> Tuple[int, int]:
        out = np.zeros(2, dtype=np.uint64)
        check(self._L.bk_rng_state(self._h, book, _lib.p64(out)))
        return int(out[0]), int(out[1])

    def live_orders(self, book: int) -> np.ndarray:
        cap = 1024
        a = np.zeros(cap, dtype=_lib.ORDER_DTYPE)
        n = C.c_uint32(0)
        check(self._L.bk_live_orders(self._h, book, cap, a.ctypes.data_as(C.c_void_p), C.byref(n)))
        return a[: n.value].copy()

    def stats(self) -> dict:
        s = Stats()
        check(self._L.bk_stats_compute(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def stats_device_ptr(self) -> int:
        out = C.c_void_p()
        check(self._L.bk_stats_device_ptr(self._h, C.byref(out)))
        return int(out.value)

    def level2_device_ptr(self) -> int:
        """Device address of the latest level-2 records, u32[n_books][width] (for on-device consumers)."""
        out = C.c_void_p()
        check(self._L.bk_level2_device_ptr(self._h, C.byref(out)))
        return int(out.value)

    def stats_compute_async(self):
        check(self._L.bk_stats_compute(self._h, None))

    # ------------------------------------------------------------ measurement
    def profile(self, every: int):
        """HIP-event timing of the step kernels: 0 off, N >= 1 = time the kernels of every Nth step."""
        check(self._L.bk_profile_enable(self._h, int(every)))

    def profile_read(self, reset: bool = True) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self._L.bk_profile_read(self._h, C.byref(ms), C.byref(n), int(reset)))
        return float(ms.value), int(n.value)

    def profile_read_kind(self, kind: int) -> Tuple[float, int]:
        """kind: 0 k_run_random, 1 k_agents_fsm, 2 k_step_batch, 3 k_step_events (call before profile_read(reset))."""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self._L.bk_profile_read_kind(self._h, int(kind), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def set_pipeline(self, mode: str):
        """'auto' | 'fused' | 'split' | 'split_wave' | 'wave_split' | 'wave' — kernel pipeline of run(); results are
        identical.  ('wave' / 'wave_split': RandomAgents books, the RNG-serial phases one wave per book with a
        wave-parallel stream decode, fused with the event phase in one persistent kernel / as a kernel of its own;
        'wave_split' on an AgentSet of Noise / Momentum members: their update one wave per book with the same kind of
        decode - k_agents_mixed_wave - in front of the event kernel, the auto choice from 512 books.)  ('split_wave':
        AgentSets with Noise/Momentum members keep their update one wave per book as scalar code; for RandomAgents it
        equals 'split'.)"""
        check(self._L.bk_set_pipeline(self._h, {"auto": 0, "fused": 1, "split": 2, "split_wave": 3, "wave_split": 4, "wave": 5}[mode]))

    def set_wave_options(self, lookahead: int = 64, parts: int = 0):
        """'wave' pipeline knobs: look-ahead of the vector decode path (1..64; small = exercise the scalar slow path)
        and the number of parts the batch is cut in (0 = default)."""
        check(self._L.bk_set_wave_options(self._h, int(lookahead), int(parts)))

    def set_split_parts(self, n_parts: int, min_part: int = 4096):
        """Split pipeline: cut the batch in ``min(n_parts, books / min_part)`` parts on separate streams."""
        check(self._L.bk_set_split_parts(self._h, int(n_parts), int(min_part)))

    def split_parts(self) -> Tuple[int, int]:
        """(n_parts, min_part) as set (``bk_get_split_parts``); ``pipeline()`` reports the effective number of parts."""
        a, b = C.c_int(0), C.c_uint32(0)
        check(self._L.bk_get_split_parts(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def pipeline_fallbacks(self) -> int:
        """Rounds 1-2: bk_run launches the auto pipeline rolled back and redid on the fused kernel.  Always 0 since round
        3 (the auto choice frees pool slots like the fused kernel; nothing is left to roll back)."""
        out = C.c_uint64(0)
        check(self._L.bk_pipeline_fallbacks(self._h, C.byref(out)))
        return int(out.value)

    def pipeline(self) -> Tuple[str, int]:
        """('fused' | 'split' | 'wave_split' | 'wave', number of book parts on separate streams) that run() will use."""
        a, b = C.c_int(0), C.c_int(1)
        check(self._L.bk_get_pipeline(self._h, C.byref(a), C.byref(b)))
        return ("fused", "split", "wave_split", "wave")[a.value], int(b.value)

    def checkpoint(self) -> np.ndarray:
        """Complete simulation state (pool, clock, counters, RNG of every book) as a byte array."""
        n = int(self._L.bk_checkpoint_bytes(self._h))
        # The flag bits a strict check has already reported live on the host (raise_on_flags moves them off the device):
        # they travel as a trailer behind the library's image, so that flags() reads the same after a restore.
        tail = 0 if self._flags_sticky is None else 16 + 4 * self.n_books
        buf = np.zeros(n + tail, dtype=np.uint8)
        check(self._L.bk_checkpoint_save(self._h, buf.ctypes.data_as(C.c_void_p), n))
        if tail:
            buf[n:n + 16] = np.frombuffer(_STICKY_MAGIC + np.uint64(self.n_books).tobytes(), dtype=np.uint8)
            buf[n + 16:] = self._flags_sticky.view(np.uint8)
        return buf

    def restore(self, buf: np.ndarray):
        """Load a checkpoint taken from an env of the same shape; the run continues bit-identically."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        n = int(self._L.bk_checkpoint_bytes(self._h))
        # exactly the library's image, or the image + the sticky-flags trailer of checkpoint() with its magic and this env's
        # book count; anything else - a checkpoint of another shape, a truncated or padded file - is refused, not loaded in part
        tail = 16 + 4 * self.n_books
        has_tail = buf.nbytes == n + tail
        if has_tail and not (buf[n:n + 8].tobytes() == _STICKY_MAGIC and
                             int(np.frombuffer(buf[n + 8:n + 16].tobytes(), dtype=np.uint64)[0]) == self.n_books):
            raise _lib.BourseError(_lib.BK_INVALID, "restore: the checkpoint's trailer is not this env's (magic / book count)")
        if not has_tail and buf.nbytes != n:
            raise _lib.BourseError(_lib.BK_INVALID, f"restore: {buf.nbytes} bytes is not a checkpoint of this env's shape ({n} bytes, "
                                                   f"or {n + tail} with the reported-flags trailer)")
        check(self._L.bk_checkpoint_load(self._h, buf.ctypes.data_as(C.c_void_p), n))
        self._flags_sticky, self._warned_bits = None, 0
        if has_tail:
            self._flags_sticky = buf[n + 16:].view(np.uint32).copy()

    def state_bytes_per_book(self) -> int:
        return int(self._L.bk_state_bytes_per_book(self._h))


def sim_runner(env: ManyBookEnv, agents: Sequence[RandomAgents | tuple], n_steps: int):
    """``sim_runner(env, agents, seed, n_steps, _)`` (ref runner.rs:46-69) for every book of ``env``.

    The seed is the env's (book b: seed + book_offset + b): in the reference the RNG is a local of
    ``sim_runner``; here it is part of the device state so runs can be continued."""
    env.set_random_agents(agents)
    env.run(n_steps)


class ManyMarketEnv(ManyBookEnv):
    """``n_markets`` independent ``MarketEnv<ASSETS>`` (ref crates/step_sim/src/market_env.rs:46-340) in lockstep: the
    ``len(tick_sizes)`` books of a market share one clock, one RNG stream (market m: seed + market_offset + m) and one
    shuffled event queue.  Every ``ManyBookEnv`` reader works on the flat book index ``market * assets + asset``
    (``book()``); the mutators below take ``(market, asset)`` like ``MarketEnv``'s take ``asset`` / ``MarketOrderId``."""

    def __init__(self, n_markets: int, seed: int, start_time: int, tick_sizes: Sequence[int], step_size: int,
                 trading: bool = True, market_offset: int = 0, **kw):
        tick_sizes = [int(t) for t in tick_sizes]
        super().__init__(int(n_markets) * len(tick_sizes), seed, start_time, tick_sizes[0], step_size, trading,
                         book_offset=market_offset, assets=len(tick_sizes), tick_sizes=tick_sizes, **kw)
        self.n_markets, self.tick_sizes = int(n_markets), tick_sizes

    def book(self, market: int, asset: int) -> int:
        if not (0 <= market < self.n_markets and 0 <= asset < self.assets):
            raise IndexError("market / asset out of range")
        return market * self.assets + asset

    def place_order(self, market: int, asset: int, bid: bool, vol: int, trader_id: int, price: Optional[int] = None):
        """``MarketEnv::place_order(asset, side, vol, trader_id, price)`` (market_env.rs:163-176) -> per-asset order id"""
        return super().place_order(self.book(market, asset), bid, vol, trader_id, price)

    def cancel_order(self, market: int, asset: int, order_id: int):
        super().cancel_order(self.book(market, asset), order_id)

    def modify_order(self, market: int, asset: int, order_id: int, new_price: Optional[int] = None,
                     new_vol: Optional[int] = None):
        super().modify_order(self.book(market, asset), order_id, new_price, new_vol)


    # Market::save_json / load_json (market.rs:367-390): {"order_books": [OrderBook; ASSETS]}
    def market_state(self, market: int, trading: bool = True) -> dict:
        return {"order_books": [self.book_state(self.book(market, a), trading) for a in range(self.assets)]}

    def load_market_state(self, market: int, state: dict):
        if len(state["order_books"]) != self.assets:
            raise ValueError("snapshot holds a different number of assets")
        for a, s in enumerate(state["order_books"]):
            self.load_book_state(self.book(market, a), s)

    def save_json(self, market: int, path: str, pretty: bool = False, trading: bool = True):
        import json

        with open(path, "w") as f:
            json.dump(self.market_state(market, trading), f, **({"indent": 2} if pretty else {"separators": (",", ":")}))

    def load_json(self, market: int, path: str):
        import json

        with open(path) as f:
            self.load_market_state(market, json.load(f))


def market_sim_runner(env: ManyMarketEnv, agents: Sequence[RandomMarketAgents | tuple], n_steps: int):
    """``market_sim_runner(env, agents, seed, n_steps, _)`` (ref runner.rs:108-131) for every market of ``env``."""
    env.set_random_market_agents(agents)
    env.run(n_steps)
