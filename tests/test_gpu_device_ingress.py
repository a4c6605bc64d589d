"""Device-resident instruction ingress: `bk_submit_instructions_device` takes the six SoA arrays of
`submit_instructions` (ref rust/src/step_sim_numpy.rs:233-275) as DEVICE pointers for every book at once - tick check,
dense per-book id assignment by a prefix sum on the GPU, the reference's partial-application semantics per book - and
feeds `Env::step` (ref crates/step_sim/src/env.rs:116-135,166-219) without the host half of `Env`.

Checked three ways on the same random instruction stream: the device entry == the per-book host calls == the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U64MAX = 2**64 - 1


@pytest.fixture(scope="module")
def bk():
    import bourse_amd

    return bourse_amd


MOD = 0x80000003  # BK_ACTION_MODIFY (include/bourse_amd.h): the one extension of the reference's action codes


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream_step(rng, B, counts, max_per_book, tick, bad_books=()):
    """One step's ragged SoA batch for B books: action 0 / 1 (new) / 2 (cancel) / BK_ACTION_MODIFY, and 3 / 7 - which the
    reference's submit_instructions ignores (step_sim_numpy.rs:266) and so must every entry here -, targets among the ids the
    book has created so far (so they hit live, filled and cancelled orders alike); `bad_books`: an odd price is planted
    in the middle of those books' batches."""
    n_b = rng.integers(0, max_per_book + 1, size=B)
    n_b[rng.random(B) < 0.05] = 0  # some books sit a step out
    off = np.zeros(B + 1, dtype=np.uint64)
    off[1:] = np.cumsum(n_b)
    n = int(off[-1])
    book_of = np.repeat(np.arange(B), n_b)
    action = rng.choice([0, 1, 2, MOD, 3, 7], size=n, p=[0.03, 0.6, 0.2, 0.15, 0.01, 0.01]).astype(np.uint32)
    have = counts[book_of]
    action[(action >= 2) & (have == 0)] = 1  # nothing to cancel / modify yet
    bid = rng.integers(0, 2, size=n).astype(np.uint8)
    has_p, has_v = rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(0, 2, size=n).astype(np.uint8)
    side = np.where(action == MOD, (has_p << 1) | (has_v << 2), bid).astype(np.uint8)
    vol = rng.integers(1, 30, size=n).astype(np.uint32)
    trader = rng.integers(0, 1000, size=n).astype(np.uint32)
    price = (rng.integers(45, 56, size=n) * tick).astype(np.uint32)
    target = (rng.random(n) * np.maximum(have, 1)).astype(np.uint64)
    order_id = np.where(action >= 2, target, 0).astype(np.uint64)
    for b in bad_books:
        lo, hi = int(off[b]), int(off[b + 1])
        news = [i for i in range(lo, hi) if action[i] == 1]
        if news:
            price[news[len(news) // 2]] += 1  # not a multiple of the tick size
    return off, book_of, (action, side, vol, trader, price, order_id)


def _apply_host(env, book_of, off, ins, B):
    """The same batch through the per-order host entries, book by book, with the reference's stop-at-first-error rule.
    Returns (ids per element, elements applied per book, error code per book)."""
    action, side, vol, trader, price, order_id = ins
    ids = np.full(len(action), U64MAX, dtype=np.uint64)
    applied, code = np.zeros(B, dtype=np.uint32), np.zeros(B, dtype=np.uint32)
    for b in range(B):
        for i in range(int(off[b]), int(off[b + 1])):
            a = int(action[i])
            try:
                if a == 1:
                    ids[i] = env.place_order(b, bool(side[i] & 1), int(vol[i]), int(trader[i]), int(price[i]))
                elif a == 2:
                    env.cancel_order(b, int(order_id[i]))
                elif a == MOD:
                    env.modify_order(b, int(order_id[i]), int(price[i]) if side[i] & 2 else None, int(vol[i]) if side[i] & 4 else None)
            except ValueError:
                code[b] = 1
                break
            applied[b] += 1
    return ids, applied, code


def _apply_oracle(ref, lo, hi, ins):
    action, side, vol, trader, price, order_id = ins
    for i in range(lo, hi):
        a = int(action[i])
        try:
            if a == 1:
                ref.place_order(bool(side[i] & 1), int(vol[i]), int(trader[i]), price=int(price[i]))
            elif a == 2:
                ref.cancel_order(int(order_id[i]))
            elif a == MOD:
                ref.modify_order(int(order_id[i]), new_price=int(price[i]) if side[i] & 2 else None,
                                 new_vol=int(vol[i]) if side[i] & 4 else None)
        except ValueError:
            return


def test_device_ingress_equals_host_calls_and_oracle_on_8192_books(bk, oracle):
    import torch

    B, T, TICK, NMAX = 8192, 5, 2, 14
    kw = dict(levels=10, max_live_orders=128, max_orders=NMAX * T + 8, trade_capacity=NMAX * T * 2, history_capacity=T)
    dev = bk.ManyBookEnv(B, 77, 0, TICK, 100_000, stream=torch.cuda.current_stream().cuda_stream, **kw)
    host = bk.ManyBookEnv(B, 77, 0, TICK, 100_000, **kw)
    dev.enable_device_ingress(queue_capacity=NMAX)
    sample = sorted(set(range(0, B, 67)) | {5, 4097, B - 1})
    bad_plan = {1: (5, 4097), 3: (B - 1, 134)}  # step -> books whose batch carries a bad price
    refs = {b: oracle.StepEnv(77 + b, 0, TICK, 100_000) for b in sample}
    rng = np.random.default_rng(2024)
    counts = np.zeros(B, dtype=np.int64)
    with pytest.raises(bk.BourseError, match="device memory"):
        dev.place_order(0, True, 1, 0, 100)  # one flow per env
    for s in range(T):
        off, book_of, ins = _stream_step(rng, B, counts, NMAX, TICK, bad_plan.get(s, ()))
        n = len(ins[0])
        d_ins = [_dev(torch, x) for x in ins]
        out_ids = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        status = torch.full((B, 2), 99, dtype=torch.int32, device="cuda")
        dev.submit_instructions_device(_dev(torch, off.astype(np.int64)), *d_ins, out_ids=out_ids, status=status)
        want_ids, want_applied, want_code = _apply_host(host, book_of, off, ins, B)
        dev.step(sync=False)
        host.step()
        got_ids = out_ids.cpu().numpy().view(np.uint64)
        st = status.cpu().numpy().view(np.uint32)
        assert np.array_equal(st[:, 0], want_code), (s, np.nonzero(st[:, 0] != want_code)[0][:5])
        assert np.array_equal(st[:, 1], want_applied), s
        bad = set(int(b) for b in np.nonzero(want_code)[0])
        assert bad <= set(bad_plan.get(s, ())) and (bool(bad) or s not in bad_plan)  # (a planted book may hold no new order)
        assert np.array_equal(got_ids, want_ids), s  # untouched (still -1 = u64::MAX) from a failing element on
        for b in sample:
            _apply_oracle(refs[b], int(off[b]), int(off[b + 1]), ins)
            refs[b].step()
        counts += np.bincount(book_of[want_ids != U64MAX], minlength=B)
    dev.sync()
    assert not dev.flags().any() and not host.flags().any()
    hd, hh = dev.history(), host.history()
    assert np.array_equal(hd, hh)
    assert np.array_equal(dev.trade_counts(), host.trade_counts()) and int(host.trade_counts().sum()) > 10_000
    assert [dev.order_count(b) for b in sample] == [int(counts[b]) for b in sample]
    for b in sample:
        assert np.array_equal(hd[:, b], refs[b].history()), b
        gd, gh, e = dev.trades(b, first=0), host.trades(b, first=0), refs[b].book.trades_array()
        od, oh, eo = dev.orders(b), host.orders(b), refs[b].book.orders_array()
        for f in gd.dtype.names:
            assert np.array_equal(gd[f], e[f]) and np.array_equal(gh[f], e[f]), (b, f)
        for f in od.dtype.names:
            assert np.array_equal(od[f], eo[f]) and np.array_equal(oh[f], eo[f]), (b, f)
        kd, kh = dev.order_keys(b), host.order_keys(b)
        assert np.array_equal(kd[0], kh[0]) and np.array_equal(kd[1], kh[1]), b
    # a second submit before the step appends (one submit per agent and step, ref src/bourse/step_sim/runner.py:108-112)
    off = np.arange(B + 1, dtype=np.int64) * 2
    one = lambda v, dt: _dev(torch, np.full(2 * B, v, dtype=dt))  # noqa: E731
    for k in range(2):
        dev.submit_instructions_device(_dev(torch, off), one(1, np.uint32), one(k, np.uint8), one(3, np.uint32), one(k, np.uint32),
                                       one(100, np.uint32), one(0, np.uint64))
    host.submit_instructions_all(off.astype(np.uint64), (np.ones(2 * B, np.uint32), np.zeros(2 * B, bool), np.full(2 * B, 3, np.uint32),
                                                         np.zeros(2 * B, np.uint32), np.full(2 * B, 100, np.uint32), np.zeros(2 * B, np.uint64)))
    host.submit_instructions_all(off.astype(np.uint64), (np.ones(2 * B, np.uint32), np.ones(2 * B, bool), np.full(2 * B, 3, np.uint32),
                                                         np.ones(2 * B, np.uint32), np.full(2 * B, 100, np.uint32), np.zeros(2 * B, np.uint64)))
    dev.step()
    host.step()
    assert np.array_equal(dev.level2(), host.level2())
    assert np.array_equal(dev.trade_counts(), host.trade_counts())
    dev.close()
    host.close()


def test_host_arrays_through_device_ingress_equal_host_calls_and_oracle_on_8192_books(bk, oracle):
    """a BaseNumpyAgent-style caller hands over HOST numpy arrays (base_agent.py:67-116, runner.py:103-112).
    The same three-way check as above - device-ingress env == per-order host entries == oracle - with the arrays taken from
    host memory by bk_submit_instructions_host: synchronously (ids returned, the reference's ValueError raised), as tickets
    with two submits in flight, and filled in place in the library's pinned staging arrays."""
    B, T, TICK, NMAX = 8192, 5, 2, 14
    kw = dict(levels=10, max_live_orders=128, max_orders=NMAX * T + 8, trade_capacity=NMAX * T * 2, history_capacity=T)
    devs = {k: bk.ManyBookEnv(B, 77, 0, TICK, 100_000, **kw) for k in ("sync", "tickets", "staging")}
    for d in devs.values():
        d.enable_device_ingress(queue_capacity=NMAX)
    host = bk.ManyBookEnv(B, 77, 0, TICK, 100_000, **kw)
    sample = sorted(set(range(0, B, 67)) | {5, 4097, B - 1})
    bad_plan = {1: (5, 4097), 3: (B - 1, 134)}
    refs = {b: oracle.StepEnv(77 + b, 0, TICK, 100_000) for b in sample}
    rng = np.random.default_rng(2025)
    counts = np.zeros(B, dtype=np.int64)
    pending = None  # (ticket, want_ids, want_applied, want_code) of the previous step on the "tickets" env

    def check(res, want, s):
        ids, st, bad = res
        want_ids, want_applied, want_code = want
        assert np.array_equal(st[:, 0], want_code) and np.array_equal(st[:, 1], want_applied), s
        # ids from a book's failing element on read u64::MAX (the device entry leaves them untouched; the per-order host
        # calls never reach them)
        assert np.array_equal(ids, want_ids), s
        assert bad == (int(np.nonzero(want_code)[0][0]) if want_code.any() else None), s

    for s in range(T):
        off, book_of, ins = _stream_step(rng, B, counts, NMAX, TICK, bad_plan.get(s, ()))
        n = len(ins[0])
        want = _apply_host(host, book_of, off, ins, B)
        # (1) synchronous: ids returned / ValueError for the first failing book, every other book applied all the same
        if want[2].any():
            with pytest.raises(ValueError, match=f"book {int(np.nonzero(want[2])[0][0])}: a price of its batch"):
                devs["sync"].submit_instructions_all(off, ins)
        else:
            assert np.array_equal(devs["sync"].submit_instructions_all(off, ins), want[0]), s
        # (2) tickets: this step's submit goes out BEFORE the previous step's results are fetched
        t = devs["tickets"].submit_instructions_all_async(off, ins)
        if pending is not None:  # (alternately copied out and as read-only views of the pinned staging)
            res = devs["tickets"].submit_result(pending[0], view=bool(s & 1))
            if s & 1:
                assert not res[0].flags.writeable and not res[1].flags.writeable
            check(res, pending[1], s - 1)
        pending = (t, want)
        # (3) the arrays written in place into the pinned staging of the next submit
        st = devs["staging"].ingress_staging(n)
        assert st["capacity"] >= n
        st["book_offsets"][:] = off
        for k, a in zip(("action", "side", "vol", "trader_id", "price", "order_id"), ins):
            st[k][:n] = a
        t3 = devs["staging"].submit_instructions_all_async(st["book_offsets"], tuple(st[k][:n] for k in ("action", "side", "vol", "trader_id", "price", "order_id")))
        check(devs["staging"].submit_result(t3), want, s)
        for d in devs.values():
            d.step(sync=False)
        host.step()
        for b in sample:
            _apply_oracle(refs[b], int(off[b]), int(off[b + 1]), ins)
            refs[b].step()
        counts += np.bincount(book_of[want[0] != U64MAX], minlength=B)
    check(devs["tickets"].submit_result(pending[0]), pending[1], T - 1)
    with pytest.raises(bk.BourseError, match="expired"):
        devs["tickets"].submit_result(pending[0] - 2)
    hh = host.history()
    assert int(host.trade_counts().sum()) > 10_000
    for name, d in devs.items():
        d.sync()
        assert not d.flags().any(), name
        hd = d.history()
        assert np.array_equal(hd, hh), name
        assert np.array_equal(d.trade_counts(), host.trade_counts()), name
        for b in sample:
            assert np.array_equal(hd[:, b], refs[b].history()), (name, b)
            gd, e = d.trades(b, first=0), refs[b].book.trades_array()
            od, eo = d.orders(b), refs[b].book.orders_array()
            for f in gd.dtype.names:
                assert np.array_equal(gd[f], e[f]), (name, b, f)
            for f in od.dtype.names:
                assert np.array_equal(od[f], eo[f]), (name, b, f)
    # an empty step (no element for any book) and a growing batch (the staging is re-allocated mid-flight)
    d = devs["tickets"]
    t0 = d.submit_instructions_all_async(np.zeros(B + 1, np.uint64), tuple(np.zeros(0, dt) for dt in (np.uint32, np.uint8, np.uint32, np.uint32, np.uint32, np.uint64)))
    ids, st, bad = d.submit_result(t0)
    assert len(ids) == 0 and not st.any() and bad is None
    big = 40 * B
    offb = np.arange(B + 1, dtype=np.uint64) * 40
    dq = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=10, max_live_orders=64, max_orders=64, trade_capacity=64)
    dq.enable_device_ingress(64)
    small = dq.submit_instructions_all(np.arange(B + 1, dtype=np.uint64), (np.ones(B, np.uint32), np.zeros(B, np.uint8), np.ones(B, np.uint32),
                                                                       np.zeros(B, np.uint32), np.full(B, 90, np.uint32), np.zeros(B, np.uint64)))
    assert (small == 0).all()
    ids = dq.submit_instructions_all(offb, (np.ones(big, np.uint32), np.ones(big, np.uint8), np.ones(big, np.uint32), np.zeros(big, np.uint32),
                                            np.full(big, 80, np.uint32), np.zeros(big, np.uint64)))
    assert np.array_equal(ids.reshape(B, 40), np.tile(np.arange(1, 41, dtype=np.uint64), (B, 1)))
    dq.step()
    assert (dq.level2()[:, 1] == 80).all() and (dq.level2()[:, 2] == 90).all() and not dq.flags().any()
    # ... and a ticket whose results have not been fetched yet survives the re-allocation a LATER, larger batch causes
    # (scripts/fuzz_device_ingress.py seed 3529, round 5: "unknown or expired ticket")
    dg = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=10, max_live_orders=128, max_orders=128, trade_capacity=64)
    dg.enable_device_ingress(128)
    one = lambda n, v, dt: np.full(n, v, dtype=dt)  # noqa: E731
    ta = dg.submit_instructions_all_async(np.arange(B + 1, dtype=np.uint64), (one(B, 1, np.uint32), one(B, 0, np.uint8), one(B, 1, np.uint32),
                                                                           one(B, 0, np.uint32), one(B, 90, np.uint32), one(B, 0, np.uint64)))
    tb = dg.submit_instructions_all_async(offb, (one(big, 1, np.uint32), one(big, 1, np.uint8), one(big, 1, np.uint32), one(big, 0, np.uint32),
                                                 one(big, 80, np.uint32), one(big, 0, np.uint64)))
    ids_a, st_a, bad_a = dg.submit_result(ta)
    ids_b, st_b, bad_b = dg.submit_result(tb, view=True)
    assert (ids_a == 0).all() and bad_a is None and (st_a[:, 1] == 1).all()
    assert np.array_equal(ids_b.reshape(B, 40), np.tile(np.arange(1, 41, dtype=np.uint64), (B, 1))) and (st_b[:, 1] == 40).all()
    # ... and a VIEW taken before a growth stays readable until two more submits (the growth freed the pinned block
    # under it): view of ticket tc, then a batch that outgrows the staging, then read the view
    dg.step()
    tc = dg.submit_instructions_all_async(np.arange(B + 1, dtype=np.uint64), (one(B, 1, np.uint32), one(B, 0, np.uint8), one(B, 2, np.uint32),
                                                                           one(B, 0, np.uint32), one(B, 95, np.uint32), one(B, 0, np.uint64)))
    ids_c, st_c, bad_c = dg.submit_result(tc, view=True)
    want_c = ids_c.copy()
    assert (want_c == 41).all() and bad_c is None
    huge = 120 * B  # 3 x the staging's capacity: re-allocated
    offh = np.arange(B + 1, dtype=np.uint64) * 120
    td = dg.submit_instructions_all_async(offh, (one(huge, 0, np.uint32), one(huge, 1, np.uint8), one(huge, 1, np.uint32), one(huge, 0, np.uint32),
                                                 one(huge, 80, np.uint32), one(huge, 0, np.uint64)))  # (null instructions: nothing queued)
    assert np.array_equal(ids_c, want_c) and (st_c[:, 1] == 1).all(), "a view of ticket t must survive the growth caused by ticket t + 1"
    ids_d, st_d, bad_d = dg.submit_result(td)
    assert (ids_d == 2**64 - 1).all() and bad_d is None
    assert np.array_equal(ids_c, want_c)
    dg.close()
    dq.close()
    for d in devs.values():
        d.close()
    host.close()


class _InsideTouch:
    """A one-book numpy agent that READS its level-2 record: one order a tick inside the touch it sees (so the instruction
    stream depends on what the books did in the step before)."""

    def __init__(self, trader):
        self.trader = trader

    def update(self, rng, l2):
        bid, ask = int(l2[1]), int(l2[2])
        side = bool(rng.integers(0, 2))
        px = (bid + 1 if bid > 0 else 40) if side else (ask - 1 if ask < 2**32 - 1 else 60)
        return (np.array([1], np.uint32), np.array([side]), np.array([int(rng.integers(1, 9))], np.uint32),
                np.array([self.trader], np.uint32), np.array([max(1, min(px, 200))], np.uint32), np.array([0], np.uint64))


@pytest.mark.parametrize("ingress", [True, False])
def test_run_many_numpy_agents_on_many_books_equals_one_oracle_env_per_book(bk, oracle, ingress):
    """bourse_amd.step_sim.run_many = the reference's numpy loop (runner.py:103-112) over a ManyBookEnv: one agent set, each
    agent's instructions for ALL books in one CSR batch of host arrays.  Against B oracle StepEnvNumpy envs driven by the
    same loop book by book (same generator, same call order), on the device-ingress flow and on the host Env half."""
    from bourse_amd.step_sim import agents as A

    B, T, SEED = 96, 12, 7
    mk = lambda: [A.NumpyRandomAgents(6, (40, 60), (1, 9), 1), _InsideTouch(99), A.NumpyRandomAgents(3, (45, 55), (1, 9), 1)]  # noqa: E731
    env = bk.ManyBookEnv(B, 21, 0, 1, 100_000, levels=10, max_live_orders=192, max_orders=16 * T, trade_capacity=32 * T,
                         history_capacity=T)
    if ingress:
        env.enable_device_ingress(16)
    hist = bk.step_sim.run_many(env, mk(), T, SEED)
    assert hist.shape == (T, B, 45)
    refs = [oracle.StepEnvNumpy(21 + b, 0, 1, 100_000) for b in range(B)]
    rng = np.random.default_rng(SEED)
    agents = mk()
    for _ in range(T):
        l2 = [r.level_2_data() for r in refs]
        for a in agents:
            for b, r in enumerate(refs):
                r.submit_instructions(a.update(rng, l2[b]))
        for r in refs:
            r.step()
    for b, r in enumerate(refs):
        assert np.array_equal(hist[:, b], r.history()), b
        got, exp = env.trades(b, first=0), r.book.trades_array()
        for f in got.dtype.names:
            assert np.array_equal(got[f], exp[f]), (b, f)
    assert int(env.trade_counts().sum()) > 1000
    # the books-vectorised agent: one update_many call per step, 8 192 books
    B2 = 8192
    big = bk.ManyBookEnv(B2, 5, 0, 2, 100_000, levels=10, max_live_orders=256, max_orders=0, trade_capacity=64, history_capacity=0, strict=False)
    if ingress:
        big.enable_device_ingress(16)
    last = bk.step_sim.run_many(big, [A.ManyBookNumpyRandomAgents(12, (40, 60), (1, 9), 2)], 4, 3)
    assert last.shape == (B2, 45) and (last[:, 1] < last[:, 2]).all() and int(big.trade_counts().sum()) > B2 and not big.flags().any()
    big.close()
    env.close()


def test_device_ingress_capacity_unknown_ids_and_argument_checks(bk):
    import torch

    B = 4
    dev = bk.ManyBookEnv(B, 3, 0, 1, 1000, levels=10, max_live_orders=64, max_orders=64, trade_capacity=64, history_capacity=8,
                         stream=torch.cuda.current_stream().cuda_stream, strict=False)
    with pytest.raises(bk.BourseError, match="bk_device_ingress_enable"):
        dev.step(sync=False)
    with pytest.raises(bk.BourseError, match="1..8192"):
        dev.enable_device_ingress(0)
    dev.enable_device_ingress(6)
    dev.enable_device_ingress(6)  # idempotent
    with pytest.raises(bk.BourseError, match="another queue capacity"):
        dev.enable_device_ingress(7)
    with pytest.raises(ValueError, match="CUDA tensor"):
        dev.submit_instructions_device(torch.zeros(B + 1, dtype=torch.int64), *[None] * 6)
    # book 0: 4 new orders; book 1: 8 (queue capacity 6 -> two dropped, BK_CAPACITY); book 2: nothing; book 3: a cancel of an id
    # that was never created between two new orders
    off = np.array([0, 4, 12, 12, 15], dtype=np.int64)
    n = 15
    action = np.ones(n, dtype=np.uint32)
    action[13] = 2
    order_id = np.zeros(n, dtype=np.uint64)
    order_id[13] = 40
    side = (np.arange(n) % 2).astype(np.uint8)
    args = [_dev(torch, x) for x in (off, action, side, np.full(n, 2, np.uint32), np.arange(n, dtype=np.uint32),
                                     np.where(side == 1, 99, 101).astype(np.uint32), order_id)]
    out_ids = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    status = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    with pytest.raises(bk.CapacityError, match="book 1"):
        dev.submit_instructions_device(*args, out_ids=out_ids, status=status, check_status=True)
    st = status.cpu().numpy()
    assert st.tolist() == [[0, 4], [3, 6], [0, 0], [0, 3]]
    ids = out_ids.cpu().numpy().view(np.uint64)
    assert ids[:4].tolist() == [0, 1, 2, 3] and ids[4:10].tolist() == [0, 1, 2, 3, 4, 5] and (ids[10:12] == U64MAX).all()
    assert ids[12] == 0 and ids[13] == U64MAX and ids[14] == 1
    dev.step()
    f = dev.flags()
    assert f.tolist() == [0, 0, 0, 16]  # BK_FLAG_UNKNOWN_ORDER on book 3 only; its two orders rest
    assert [dev.order_count(b) for b in range(B)] == [4, 6, 0, 2]
    assert len(dev.live_orders(3)) == 2 and len(dev.live_orders(1)) == 6
    # the same id once it exists (a filled / cancelled / live order): no flag
    dev.clear_flags()
    off2 = np.array([0, 0, 0, 0, 1], dtype=np.int64)
    a2 = [_dev(torch, x) for x in (off2, np.array([2], np.uint32), np.zeros(1, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.uint32),
                                   np.zeros(1, np.uint32), np.array([1], np.uint64))]
    dev.submit_instructions_device(*a2)
    dev.step()
    assert not dev.flags().any() and len(dev.live_orders(3)) == 1
    with pytest.raises(bk.BourseError):
        dev.run(1)  # one flow per env
    dev.close()


def test_device_ingress_markets_match_host_calls(bk):
    """MarketEnv mode (assets = 2): the queue is the MARKET's - asset 0's batch then asset 1's, as the per-book host calls
    queue them - ids stay per book, ticks per asset."""
    import torch

    NM, M = 300, 2
    kw = dict(levels=10, max_live_orders=64, max_orders=80, trade_capacity=128, history_capacity=4)
    dev = bk.ManyMarketEnv(NM, 9, 0, [1, 2], 100_000, stream=torch.cuda.current_stream().cuda_stream, **kw)
    host = bk.ManyMarketEnv(NM, 9, 0, [1, 2], 100_000, **kw)
    dev.enable_device_ingress(32)
    rng = np.random.default_rng(5)
    B = NM * M
    counts = np.zeros(B, dtype=np.int64)
    for s in range(4):
        off, book_of, ins = _stream_step(rng, B, counts, 9, 2, bad_books=(7,) if s == 2 else ())
        # prices are multiples of 2: fine for both assets (ticks 1 and 2); the planted odd price only offends asset 1 (book 7)
        out_ids = torch.full((len(ins[0]),), -1, dtype=torch.int64, device="cuda")
        status = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
        dev.submit_instructions_device(_dev(torch, off.astype(np.int64)), *[_dev(torch, x) for x in ins], out_ids=out_ids, status=status)
        want_ids, want_applied, want_code = _apply_host(_BookView(host), book_of, off, ins, B)
        dev.step()
        host.step()
        assert np.array_equal(out_ids.cpu().numpy().view(np.uint64), want_ids), s
        st = status.cpu().numpy().view(np.uint32)
        assert np.array_equal(st[:, 0], want_code) and np.array_equal(st[:, 1], want_applied), s
        counts += np.bincount(book_of[want_ids != U64MAX], minlength=B)
    assert np.array_equal(dev.history(), host.history())
    assert np.array_equal(dev.trade_counts(), host.trade_counts()) and int(host.trade_counts().sum()) > 100
    for b in (0, 1, 7, B - 1):
        gd, gh = dev.trades(b, first=0), host.trades(b, first=0)
        for f in gd.dtype.names:
            assert np.array_equal(gd[f], gh[f]), (b, f)


class _BookView:
    """ManyMarketEnv addressed by flat book index (its own methods take (market, asset))."""

    def __init__(self, env):
        from bourse_amd.env import ManyBookEnv

        self.env, self.base = env, ManyBookEnv

    def place_order(self, b, *a):
        return self.base.place_order(self.env, b, *a)

    def cancel_order(self, b, i):
        return self.base.cancel_order(self.env, b, i)

    def modify_order(self, b, i, p, v):
        return self.base.modify_order(self.env, b, i, p, v)


def test_two_envs_on_two_streams_equal_one_env(bk):
    """INTEGRATION.md's two-env form of the ingress: the books of one env as two envs with `book_offset` on two streams (each
    env's k_ingest under the other's k_step_events) - same seeds per global book (ref crates/step_sim/src/runner.py-style
    independence: crates/step_sim/src/runner.rs:46-69), same instructions, same results."""
    import torch

    B, H, N, T = 1024, 512, 24, 6
    kw = dict(levels=10, max_live_orders=128, max_orders=N * T + 8, trade_capacity=4 * N * T, history_capacity=T, strict=False)
    one = bk.ManyBookEnv(B, 9, 0, 1, 100_000, stream=torch.cuda.current_stream().cuda_stream, **kw)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    two = [bk.ManyBookEnv(H, 9, 0, 1, 100_000, book_offset=H * p, stream=streams[p].cuda_stream, **kw) for p in (0, 1)]
    for e in [one] + two:
        e.enable_device_ingress(N)
    rng = np.random.default_rng(3)
    off_all = _dev(torch, (np.arange(B + 1) * N).astype(np.int64))
    off_half = _dev(torch, (np.arange(H + 1) * N).astype(np.int64))
    for s in range(T):
        n = B * N
        canc = (rng.random(n) < 0.3) & (s > 0)
        ins = (np.where(canc, 2, 1).astype(np.uint32), rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(1, 30, size=n).astype(np.uint32),
               np.zeros(n, dtype=np.uint32), rng.integers(95, 106, size=n).astype(np.uint32),
               np.where(canc, rng.integers(0, max(1, s * N // 2), size=n), 0).astype(np.uint64))
        one.submit_instructions_device(off_all, *[_dev(torch, x) for x in ins])
        one.step(sync=False)
        torch.cuda.synchronize()  # (the halves' uploads below happen on the current stream)
        for p in (0, 1):
            part = [_dev(torch, x[p * H * N:(p + 1) * H * N]) for x in ins]
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[p]):
                two[p].submit_instructions_device(off_half, *part)
                two[p].step(sync=False)
    torch.cuda.synchronize()
    assert not one.flags().any() and not two[0].flags().any() and not two[1].flags().any()
    h1 = one.history()
    assert np.array_equal(h1[:, :H], two[0].history()) and np.array_equal(h1[:, H:], two[1].history())
    tc = one.trade_counts()
    assert np.array_equal(tc[:H], two[0].trade_counts()) and np.array_equal(tc[H:], two[1].trade_counts()) and int(tc.sum()) > 1000
    for b in (0, 77, H - 1):
        for p in (0, 1):
            a, c = one.trades(p * H + b, first=0), two[p].trades(b, first=0)
            assert all(np.array_equal(a[f], c[f]) for f in a.dtype.names)
    for e in [one] + two:
        e.close()


def test_the_rate_scripts_stream_at_full_size_runs_keyed_and_equals_the_oracle(bk, oracle):
    """The stream of scripts/device_ingress_rate.py / `bench.py --workload INGRESS` at its full size (8 192 books x 48 instructions
    per book-step, 70 % new limit orders, 30 % cancellations of earlier ids): every book-step on the keyed form of the host-driven
    step (step_events.hpp), every 64th book against the oracle's StepEnvNumpy fed the same arrays (native submit_instructions,
    ref rust/src/step_sim_numpy.rs:233-275; Env::step crates/step_sim/src/env.rs:116-135): level 2 of every step, trades, orders."""
    import torch

    B, N, T = 8192, 48, 10
    env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=10, max_live_orders=256, max_orders=N * T + 8, trade_capacity=64 * T, history_capacity=T,
                         strict=False, stream=torch.cuda.current_stream().cuda_stream)
    env.enable_device_ingress(N)
    sample = list(range(0, B, 64))
    refs = {b: oracle.StepEnvNumpy(1 + b, 0, 1, 100_000) for b in sample}
    rng = np.random.default_rng(0)
    off = _dev(torch, (np.arange(B + 1) * N).astype(np.int64))
    n = B * N
    for s in range(T):
        canc = (rng.random(n) < 0.3) & (s > 0)
        ins = (np.where(canc, 2, 1).astype(np.uint32), rng.integers(0, 2, size=n).astype(np.uint8), rng.integers(1, 30, size=n).astype(np.uint32),
               np.zeros(n, dtype=np.uint32), rng.integers(90, 111, size=n).astype(np.uint32),
               np.where(canc, (rng.random(n) * max(1, int(s * N * 0.6))).astype(np.uint64), 0).astype(np.uint64))
        env.submit_instructions_device(off, *[_dev(torch, x) for x in ins])
        env.step(sync=False)
        for b in sample:
            refs[b].submit_instructions_native(tuple(x[b * N:(b + 1) * N] for x in ins))
            refs[b].step()
    env.sync()
    assert not env.flags().any()
    assert np.array_equal(env.event_steps_keyed(), np.full(B, T, dtype=np.uint64)), "a book-step of this stream left the keyed form"
    h = env.history()
    n_tr = 0
    for b in sample:
        assert np.array_equal(h[:, b], refs[b].history()), b
        got, want = env.trades(b, first=0), refs[b].book.trades_array()
        assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), b
        got, want = env.orders(b), refs[b].book.orders_array()
        assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), b
        n_tr += len(want)
    assert n_tr > 100 * len(sample)
    env.close()


@pytest.mark.parametrize("pool,fracs", [(256, (0.05, 0.02)), (128, (0.05, 0.02)), (512, (0.10, 0.04)), (256, (0.0, 0.0))])
def test_the_benchs_external_agents_stream_equals_one_oracle_env_per_book(bk, oracle, pool, fracs):
    """`bench.py --workload INGRESS [--modify-frac --market-frac]` measures bench.ingress_batch's stream; this steps THAT stream -
    new limit orders, cancellations, modifications (price / volume / both), market orders - through bk_submit_instructions_device +
    bk_step_async and through one oracle StepEnvNumpy per book: ids, level 2 of every step, every trade, the whole order log.  On the
    pools of <= 256 slots every book-step must run on the keyed loop."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    B, N, T = 160, 48, 7
    n = B * N
    # (levels = 10: the oracle's StepEnvNumpy has the Python surface's fixed ladder depth; the bench's 16 only widens the record)
    env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=10, max_live_orders=pool, max_orders=N * (T + 2), trade_capacity=64 * (T + 2),
                         strict=False, history_capacity=T, stream=torch.cuda.current_stream().cuda_stream)
    env.enable_device_ingress(N)
    refs = [oracle.StepEnvNumpy(1 + b, 0, 1, 100_000) for b in range(B)]
    g = torch.Generator(device="cuda").manual_seed(3)
    off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * N
    out_ids = torch.empty(n, dtype=torch.int64, device="cuda")
    status = torch.empty((B, 2), dtype=torch.int32, device="cuda")
    dts = (np.uint32, np.uint8, np.uint32, np.uint32, np.uint32, np.uint64)
    n_mod = n_mkt = 0
    for s in range(T):
        batch = bench.ingress_batch(torch, g, n, N, s, *fracs)
        env.submit_instructions_device(off, *batch, out_ids=out_ids, status=status)
        env.step(sync=False)
        host = [np.ascontiguousarray(x.cpu().numpy()).view(dt) if x.cpu().numpy().dtype.itemsize == np.dtype(dt).itemsize
                else x.cpu().numpy().astype(dt) for x, dt in zip(batch, dts)]
        n_mod += int((host[0] == 0x80000003).sum())
        n_mkt += int(((host[0] == 1) & ((host[4] == 0) | (host[4] == 0xFFFFFFFF))).sum())
        got = out_ids.cpu().numpy().view(np.uint64)
        for b, r in enumerate(refs):
            want = r.submit_instructions_native(tuple(a[b * N:(b + 1) * N] for a in host))
            assert np.array_equal(got[b * N:(b + 1) * N], want), (s, b)
            r.step()
    env.sync()
    assert not int(status[:, 0].max()) and not env.flags().any()
    if fracs[0]:
        assert n_mod > 0.5 * fracs[0] * n * (T - 1) and n_mkt > 0.5 * fracs[1] * n * T
    h = env.history()
    for b, r in enumerate(refs):
        assert np.array_equal(h[:, b], r.history()), b
        for got, want in ((env.trades(b, first=0), r.book.trades_array()), (env.orders(b), r.book.orders_array())):
            assert len(got) == len(want), b
            for f in got.dtype.names:
                assert np.array_equal(got[f], want[f]), (b, f)
    keyed = env.event_steps_keyed()
    if pool <= 256:
        assert (keyed == T).all(), keyed[:16]
    else:  # (512 slots: the kernel with the modification code takes over a step or more after the first modification - same results)
        assert keyed.sum() >= 0.3 * B * T
    env.close()
