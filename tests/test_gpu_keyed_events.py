"""The host-driven step on the keyed loop (bourse_amd/csrc/step_events.hpp step_events_keyed, round 5; modifications and the
small pools' market orders since round 6).

`Env::step` over submitted instructions (ref crates/step_sim/src/env.rs:116-135; place / cancel / modify:
crates/order_book/src/orderbook.rs:583-611, 622-644, 743-772) runs on the slot-addressed assembly loops whenever a step
fits one event per pool slot, has no volume of 0 and its prices fit the key window - modifications included: the list is cut at
each of them and the modification (reduce in place / replace = out, re-match, rest with a fresh stamp: :656-723) happens
between two statements; everything else runs the event-by-event loop.  Both must be the reference's step, and a book must be able to alternate between them: EVERY book of a batch is compared
with its own oracle env - every step's level-2 record, every trade, the whole order log and the priority keys - on streams
that mix the two kinds of step and lean on what the keyed form re-derives instead of observing: cancellations of orders
placed in the same step (before and after their placement in the shuffled order), repeated cancellations, cancellations of
orders already filled, market orders, orders filled on arrival, partial fills followed by a cancellation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bk():
    import bourse_amd

    return bourse_amd


def _drive(bk, oracle, pool, n_max, B, T, seed, p_market, p_mod, p_zero, tick=1, lo=94, hi=107, vols=None, far=None):
    """B books x T steps of random host calls, the same calls into one oracle env per book.  Returns the env, the oracle
    envs, and per (step, book): had events / had only what the keyed form takes (as far as the calls alone can tell)."""
    env = bk.ManyBookEnv(B, seed, 0, tick, 100_000, levels=10, max_live_orders=pool, max_orders=2 * n_max * T + 8,
                         trade_capacity=6 * n_max * T + 64, history_capacity=T, strict=False)
    refs = [oracle.StepEnv(seed + b, 0, tick, 100_000) for b in range(B)]
    rng = np.random.default_rng(seed)
    made = np.zeros(B, dtype=np.int64)
    busy, clean = np.zeros((T, B), dtype=bool), np.zeros((T, B), dtype=bool)
    for s in range(T):
        for b in range(B):
            n = int(rng.integers(1, n_max + 1)) if rng.random() > 0.04 else 0
            ok, n_ev = True, 0
            for u in rng.random(n):
                if u < 0.36 and made[b] > 0:
                    # a cancellation, mostly of a recent id: live, placed in THIS step, filled, or cancelled before
                    span = made[b] if rng.random() < 0.3 else min(made[b], 3 * n_max)
                    oid = int(made[b] - 1 - rng.integers(0, span))
                    for _ in range(2 if rng.random() < 0.1 else 1):  # ... sometimes twice
                        env.cancel_order(b, oid)
                        refs[b].cancel_order(oid)
                        n_ev += 1
                elif u < 0.36 + p_mod and made[b] > 0:
                    oid = int(made[b] - 1 - rng.integers(0, min(made[b], 2 * n_max)))
                    new_p = int(rng.integers(lo, hi)) * tick if rng.random() < 0.6 else None
                    new_v = (0 if rng.random() < p_zero else int(rng.integers(1, 40))) if rng.random() < 0.6 else None
                    for _ in range(2 if rng.random() < 0.1 else 1):  # ... sometimes the same order twice in a step
                        env.modify_order(b, oid, new_p, new_v)
                        refs[b].modify_order(oid, new_p, new_v)
                        n_ev += 1
                    ok = ok and new_v != 0  # (a modification to volume 0 leaves an Active order of volume 0: event by event)
                else:
                    bid, trader = bool(rng.integers(0, 2)), int(rng.integers(0, 50))
                    vol = 0 if rng.random() < p_zero else (int(rng.integers(1, 40)) if vols is None else int(rng.choice(vols)))
                    price = None if rng.random() < p_market else int(rng.integers(lo, hi)) * tick
                    if far is not None and bid == (far[1] < lo) and price is not None and rng.random() < far[0]:
                        # a stink bid far below the book / (far range above it) a stink ask far above: the wide key windows
                        price = int(rng.integers(far[1], far[2])) * tick
                    assert env.place_order(b, bid, vol, trader, price) == made[b] == refs[b].place_order(bid, vol, trader, price)
                    made[b] += 1
                    ok, n_ev = ok and vol != 0, n_ev + 1
            busy[s, b], clean[s, b] = n_ev > 0, ok and 0 < n_ev  # (round 6: a queue longer than the pool runs keyed chunk by chunk)
        env.step()
        for r in refs:
            r.step()
    return env, refs, busy, clean


def _same_as_oracle(env, refs, allow_flags=False):
    """(allow_flags - the fuzzer's crowded shapes: a book whose pool overflowed dropped an order and is not compared)"""
    flags = env.flags()
    assert allow_flags or not flags.any(), np.unique(flags)
    h = env.history()
    for b, ref in enumerate(refs):
        if flags[b]:
            continue
        assert np.array_equal(h[:, b], ref.history()), ("level 2", b)
        got, want = env.trades(b, first=0), ref.book.trades_array()
        assert len(got) == len(want), ("trades", b, len(got), len(want))
        for f in got.dtype.names:
            assert np.array_equal(got[f], want[f]), ("trade", b, f)
        got, want = env.orders(b), ref.book.orders_array()
        assert len(got) == len(want), ("orders", b)
        for f in got.dtype.names:
            assert np.array_equal(got[f], want[f]), ("order", b, f, np.nonzero(got[f] != want[f])[0][:5])


# The step's shuffle (env.rs:121) has two forms as well: draw by draw, and - queues of at least max(32, 12 x pool registers)
# events - the decode's wave-parallel Fisher-Yates (step_events.hpp, wave_agents.hpp WaveDecoder::shuffle).  The library reads
# its two measurement knobs at every launch, so each stream runs under the shipped rule, with the wave-parallel form from two
# events on, and draw by draw.
SHUFFLES = {"shipped rule": {}, "wave-parallel from 2 events": {"BOURSE_AMD_EV_WAVE_SHUFFLE_MIN": "2"}, "draw by draw": {"BOURSE_AMD_EV_SEQ_SHUFFLE": "1"}}


@pytest.mark.parametrize("shuffle", list(SHUFFLES))
@pytest.mark.parametrize("pool,n_max,p_market", [(64, 12, 0.004), (128, 28, 0.004), (256, 44, 0.04), (512, 72, 0.04)])
def test_mixed_streams_alternate_between_the_keyed_and_the_event_by_event_loop(bk, oracle, monkeypatch, pool, n_max, p_market, shuffle):
    for k, v in SHUFFLES[shuffle].items():
        monkeypatch.setenv(k, v)
    B, T = 192, 12
    env, refs, busy, clean = _drive(bk, oracle, pool, n_max, B, T, 500 + pool, p_market, p_mod=0.05, p_zero=0.004)
    _same_as_oracle(env, refs)
    keyed = env.event_steps_keyed()
    # every keyed step was a step the keyed form may take; and it took (nearly) all of those - what the calls cannot tell is
    # a pool without a spare slot, a resting order of volume 0 and the key window
    assert np.all(keyed <= clean.sum(axis=0)), "a step with a volume of 0 ran keyed"
    assert keyed.sum() >= 0.85 * clean.sum(), (int(keyed.sum()), int(clean.sum()))
    assert 0.25 * busy.sum() < clean.sum() < busy.sum(), "the stream should mix both kinds of step"
    assert sum(len(r.book.trades_array()) for r in refs) > 20 * B
    env.close()


def test_clean_streams_run_every_step_keyed(bk, oracle):
    B, T = 256, 10
    # (up to 70 events per step on 256 slots: most steps are past the shipped rule's 48 - the wave-parallel shuffle)
    env, refs, busy, clean = _drive(bk, oracle, 256, 70, B, T, 77, p_market=0.03, p_mod=0.0, p_zero=0.0)
    _same_as_oracle(env, refs)
    assert np.array_equal(busy, clean)
    assert np.array_equal(env.event_steps_keyed(), busy.sum(axis=0))
    env.close()


def _std(i0=0, n=10):
    """n orders around 100 that cross a little: bids 98..102, asks 100..104"""
    return [("place_order", i % 2 == 0, 1 + (i0 + i) % 7, i, 100 + (i % 5) - (2 if i % 2 == 0 else 0)) for i in range(n)] + [("cancel_order", 3)]


def test_steps_outside_the_keyed_form_fall_back_and_the_books_go_on(bk, oracle):
    """One book per condition, three steps each; the book's keyed count says which loop ran, the oracle says both are right."""
    apart = [("place_order", i % 2 == 0, 5, i, (90 - i) if i % 2 == 0 else (110 + i)) for i in range(62)]  # nothing trades: 62 rest
    cases = {
        "clean": ([_std(), _std(1), _std(2)], {3}),
        "modify": ([_std(), _std(1) + [("modify_order", 1, 101, None)], _std(2)], {3}),  # (round 6: keyed)
        "modify to volume 0": ([_std(), _std(1) + [("modify_order", 1, None, 0)], _std(2)], {1, 2}),
        # (the order of volume 0 rests - orderbook.rs:430 never enters the match loop with it - and while it does, steps stay
        # on the event-by-event loop; whether step 2 finds it still there depends on the shuffle)
        "zero volume": ([_std(), _std(1) + [("place_order", True, 0, 3, 100)], _std(2)], {1, 2}),
        "more events than slots": ([_std(), [("cancel_order", 0)] * 70 + _std(1, 2), _std(2)], {3}),  # (round 6: two chunks)
        "no spare slot": ([apart, [("place_order", True, 5, 1, 20), ("place_order", False, 5, 1, 200)], [("cancel_order", 5), ("cancel_order", 63)]], {1}),  # (step 2 too: 64 live orders leave no spare slot)
        # a bid 39 900 ticks above the asks (the window spans 32 762) and an ask far below the bids: both fill on arrival
        "price window": ([_std(), _std(1) + [("place_order", True, 1, 3, 40_000), ("place_order", False, 1, 3, 1)], _std(2)], {1, 2}),
    }
    names = list(cases)
    B, T = len(names), 3
    env = bk.ManyBookEnv(B, 9, 0, 1, 100_000, levels=10, max_live_orders=64, max_orders=400, trade_capacity=400, history_capacity=T)
    refs = [oracle.StepEnv(9 + b, 0, 1, 100_000) for b in range(B)]
    for s in range(T):
        for b, name in enumerate(names):
            for f, *args in cases[name][0][s]:
                getattr(env, f)(b, *args)
                getattr(refs[b], f)(*args)
        env.step()
        for r in refs:
            r.step()
    assert not env.flags().any(), dict(zip(names, env.flags().tolist()))
    keyed = dict(zip(names, env.event_steps_keyed().tolist()))
    for name in names:
        assert keyed[name] in cases[name][1], keyed
    _same_as_oracle(env, refs)
    env.close()


def test_a_book_with_trading_disabled_steps_event_by_event(bk, oracle):
    env = bk.ManyBookEnv(2, 5, 0, 1, 100_000, levels=10, max_live_orders=128, max_orders=200, trade_capacity=200, history_capacity=4)
    refs = [oracle.StepEnv(5 + b, 0, 1, 100_000) for b in range(2)]
    for s in range(4):
        if s == 1:
            env.disable_trading()
            for r in refs:
                r.disable_trading()
        if s == 3:
            env.enable_trading()
            for r in refs:
                r.enable_trading()
        for b in range(2):
            for f, *args in _std(s) + ([("place_order", True, 4, 2, None)] if s in (1, 2) else []):  # (a market order: Rejected while trading is off)
                getattr(env, f)(b, *args)
                getattr(refs[b], f)(*args)
        env.step()
        for r in refs:
            r.step()
    assert env.event_steps_keyed().tolist() == [2, 2]
    _same_as_oracle(env, refs)
    env.close()


@pytest.mark.parametrize("assets,pool", [(2, 128), (3, 256), (4, 512)])
def test_markets_books_run_the_keyed_form_on_their_markets_queue(bk, oracle, assets, pool):
    """MarketEnv (ref crates/step_sim/src/market_env.rs:110-121): ONE shuffled queue per market, every asset's book processes its
    own events at the market's positions.  In the keyed form the other assets' events stay in a book's list as events that do
    nothing; an asset's modifications cut only THAT asset's book's list (round 6)."""
    NM, T = 48, 10
    ticks = [1, 2, 5, 1][:assets]
    env = bk.ManyMarketEnv(NM, 70, 0, ticks, 100_000, levels=10, max_live_orders=pool, max_orders=2048, trade_capacity=4096, history_capacity=T)
    ref = oracle.ManyMarkets(NM, 70, 0, ticks, 100_000, True, 10)
    rng = np.random.default_rng(assets)
    clean = np.zeros((T, NM, assets), dtype=bool)
    for s in range(T):
        for m in range(NM):
            n = int(rng.integers(1, min(60, pool // 2)))
            ok = [True] * assets
            for _ in range(n):
                a = int(rng.integers(0, assets))
                made = ref.book(m, a).n_orders()
                u = rng.random()
                if u < 0.35 and made:
                    oid = int(made - 1 - rng.integers(0, min(made, 60)))
                    env.cancel_order(m, a, oid)
                    ref.cancel_order(m, a, oid)
                elif u < 0.40 and made:
                    oid = int(made - 1 - rng.integers(0, min(made, 60)))
                    nv = int(rng.integers(1, 30))
                    np_ = int(rng.integers(95, 106)) * ticks[a] if rng.random() < 0.5 else None
                    env.modify_order(m, a, oid, np_, nv)
                    ref.modify_order(m, a, oid, np_, nv)
                else:
                    bid, vol = bool(rng.integers(0, 2)), int(rng.integers(1, 30))
                    price = None if rng.random() < 0.04 else int(rng.integers(95, 106)) * ticks[a]
                    assert env.place_order(m, a, bid, vol, 7, price) == ref.place_order(m, a, bid, vol, 7, price)
            clean[s, m] = ok
        env.step()
        ref.step()
    assert np.array_equal(env.history(), ref.history())
    for m in range(NM):
        for a in range(assets):
            b = env.book(m, a)
            got, want = env.trades(b, first=0), ref.book(m, a).trades_array()
            assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), (m, a)
            got, want = env.orders(b), ref.book(m, a).orders_array()
            assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), (m, a)
    keyed = env.event_steps_keyed().reshape(NM, assets)
    assert np.all(keyed <= clean.sum(axis=0)) and keyed.sum() >= 0.9 * clean.sum(), (int(keyed.sum()), int(clean.sum()))
    env.close()


@pytest.mark.parametrize("pool", [64, 128, 256, 512])
def test_modifications_on_the_keyed_loop_case_by_case(bk, oracle, pool):
    """orderbook.rs:743-772 through the cut list: every branch of modify_order, on orders resting from an earlier step and on
    orders of the same step (the shuffle decides which side of its placement a modification lands on - several seeds of the
    same calls), each book against its oracle env: level 2, every trade, the whole order log with the priority keys."""
    rest = [("place_order", True, 10, 1, 95), ("place_order", True, 7, 2, 96), ("place_order", False, 9, 3, 104), ("place_order", False, 6, 4, 105),
            ("place_order", True, 5, 5, 96), ("place_order", False, 4, 6, 104)]  # ids 0..5, nothing crosses; 96 and 104 hold two orders each
    cases = {
        "reduce in place keeps priority": [rest, [("modify_order", 1, None, 3), ("place_order", False, 8, 9, 96)]],
        "same volume is a replace: priority lost": [rest, [("modify_order", 1, None, 7), ("place_order", False, 8, 9, 96)]],
        "larger volume: replace": [rest, [("modify_order", 1, None, 20), ("place_order", False, 30, 9, 96)]],
        "price only, still resting": [rest, [("modify_order", 0, 97, None), ("place_order", False, 8, 9, 96)]],
        "price that crosses: re-match, remainder rests": [rest, [("modify_order", 0, 104, None)]],
        "price that crosses and fills": [rest, [("modify_order", 4, 105, 3)]],
        "ask down through the bids": [rest, [("modify_order", 3, 95, 30)]],
        "price and volume": [rest, [("modify_order", 2, 103, 2), ("modify_order", 5, 103, 2), ("place_order", True, 3, 9, 103)]],
        "neither field": [rest, [("modify_order", 2, None, None)] + _std(1, 4)],
        "twice in one step": [rest, [("modify_order", 1, 97, None), ("modify_order", 1, None, 2), ("place_order", False, 8, 9, 96)]],
        "modify then cancel then modify": [rest, [("modify_order", 1, 97, 9), ("cancel_order", 1), ("modify_order", 1, 98, 9), ("place_order", False, 8, 9, 95)]],
        "of an order of the same step": [rest, [("place_order", True, 6, 7, 97), ("modify_order", 6, 98, None), ("modify_order", 6, None, 2), ("place_order", False, 9, 8, 97)]],
        "same step, crossing after the change": [rest, [("place_order", True, 6, 7, 97), ("modify_order", 6, 104, 12), ("cancel_order", 6)]],
        "of a filled and of a cancelled order": [rest + [("place_order", False, 10, 8, 95)], [("cancel_order", 1), ("modify_order", 0, 99, 5), ("modify_order", 1, 99, 5)] + _std(1, 4)],
        "of a market order's id": [rest, [("place_order", True, 3, 7, None), ("modify_order", 6, 99, 4)] + _std(1, 4)],
        "three steps of them": [rest, [("modify_order", i, 96 + i, 5 + i) for i in range(6)], [("modify_order", i, None, 2) for i in range(6)] + _std(2, 6)],
    }
    for seed_shift in range(3):  # (the same calls under three shuffles)
        shifted = {f"{k} #{seed_shift}": [[("place_order", True, 1, 0, 50)] * 0 + step for step in v] for k, v in cases.items()}
        names = list(shifted)
        B, T = len(names), 3
        env = bk.ManyBookEnv(B, 31 + 1000 * seed_shift, 0, 1, 100_000, levels=10, max_live_orders=pool, max_orders=600, trade_capacity=800, history_capacity=T)
        refs = [oracle.StepEnv(31 + 1000 * seed_shift + b, 0, 1, 100_000) for b in range(B)]
        for s in range(T):
            for b, name in enumerate(names):
                for f, *args in (shifted[name][s] if s < len(shifted[name]) else []):
                    getattr(env, f)(b, *args)
                    getattr(refs[b], f)(*args)
            env.step()
            for r in refs:
                r.step()
        assert not env.flags().any()
        _same_as_oracle(env, refs)
        keyed = dict(zip(names, env.event_steps_keyed().tolist()))
        assert all(v == sum(1 for st in shifted[k] if st) for k, v in keyed.items()), keyed  # every step with events ran keyed
        env.close()


@pytest.mark.parametrize("pool,n_max", [(64, 14), (128, 30), (256, 48), (512, 90)])
def test_streams_with_many_modifications_and_market_orders_run_keyed(bk, oracle, pool, n_max):
    """The stream shape VERDICT r5 asked a rate for: 5 - 10 % modifications, 2 - 4 % market orders, no volume of 0 - every step with
    events must run on the keyed loop (minus the few whose pool has no spare slot) and equal the oracle."""
    B, T = 160, 12
    env, refs, busy, clean = _drive(bk, oracle, pool, n_max, B, T, 900 + pool, p_market=0.03, p_mod=0.08, p_zero=0.0)
    _same_as_oracle(env, refs)
    assert np.array_equal(busy, clean)
    keyed = env.event_steps_keyed()
    assert np.all(keyed <= busy.sum(axis=0)) and keyed.sum() >= 0.97 * busy.sum(), (int(keyed.sum()), int(busy.sum()))
    assert sum(len(r.book.trades_array()) for r in refs) > 10 * B
    env.close()


@pytest.mark.parametrize("pool,n_lo,n_hi", [(64, 70, 200), (128, 130, 420), (256, 300, 700)])
def test_queues_longer_than_the_pool_run_keyed_chunk_by_chunk(bk, oracle, pool, n_lo, n_hi):
    """More events in a step than the pool has slots (round 6): 64 R events at a time on the keyed loop, an order placed by one
    chunk resting (or gone) for the next, trade and log time stamps continuing across the chunks - against the oracle, with
    cancellations and modifications of ids of the same step (after the shuffle: in an earlier, the same or a later chunk)."""
    B, T = 96, 4
    env = bk.ManyBookEnv(B, 4242, 0, 1, 100_000, levels=10, max_live_orders=pool, max_orders=T * n_hi + 8, trade_capacity=4 * T * n_hi,
                         history_capacity=T, strict=False)
    refs = [oracle.StepEnv(4242 + b, 0, 1, 100_000) for b in range(B)]
    rng = np.random.default_rng(pool)
    made = np.zeros(B, dtype=np.int64)
    n_events = np.zeros((T, B), dtype=np.int64)
    for s in range(T):
        for b in range(B):
            n = int(rng.integers(n_lo, n_hi))
            for u in rng.random(n):
                if u < 0.40 and made[b] > 0:
                    oid = int(made[b] - 1 - rng.integers(0, min(made[b], 60)))  # mostly ids placed earlier in THIS step's calls
                    if u < 0.32:
                        env.cancel_order(b, oid)
                        refs[b].cancel_order(oid)
                    else:
                        new_p = int(rng.integers(96, 105)) if rng.random() < 0.5 else None
                        new_v = int(rng.integers(1, 9)) if (new_p is None or rng.random() < 0.5) else None
                        env.modify_order(b, oid, new_p, new_v)
                        refs[b].modify_order(oid, new_p, new_v)
                else:
                    # (tight around 100 so that most orders trade away or get cancelled: the pool must not fill up)
                    bid, vol = bool(rng.integers(0, 2)), int(rng.integers(1, 9))
                    price = None if rng.random() < 0.02 else int(rng.integers(98, 103))
                    assert env.place_order(b, bid, vol, 3, price) == made[b] == refs[b].place_order(bid, vol, 3, price)
                    made[b] += 1
                n_events[s, b] += 1
        env.step()
        for r in refs:
            r.step()
    _same_as_oracle(env, refs, allow_flags=True)
    flags = env.flags()
    assert (flags == 0).sum() >= 0.7 * B, "most books should get through without a pool overflow"
    keyed = env.event_steps_keyed()
    assert n_events.min() > pool
    # a chunk without a spare pool slot (or outside the key window) hands the rest of its step to the event-by-event loop
    assert keyed[flags == 0].sum() >= 0.6 * T * int((flags == 0).sum()), (int(keyed.sum()), T * B)
    env.close()


def test_512_slot_pools_switch_to_the_kernel_with_modifications_once_one_was_seen(bk, oracle):
    """k_step_events<8, .., MODS>: the 512-slot kernel runs WITHOUT the modification code (round 5's register budget) until its env has
    seen a modification - k_ingest's hint word for the device ingress - and with it from then on.  The hint may lag a step: that
    step's modifications run event by event.  Results equal the oracle's throughout; the keyed counter shows the switch."""
    B, N = 64, 40
    env = bk.ManyBookEnv(B, 7, 0, 1, 100_000, levels=10, max_live_orders=512, max_orders=N * 8, trade_capacity=N * 16, history_capacity=8)
    env.enable_device_ingress(N)
    refs = [oracle.StepEnvNumpy(7 + b, 0, 1, 100_000) for b in range(B)]
    rng = np.random.default_rng(5)
    off = np.arange(B + 1, dtype=np.uint64) * N
    MOD = 0x80000003
    keyed_after = []
    for s in range(6):
        n = B * N
        action = np.ones(n, np.uint32)
        side = rng.integers(0, 2, n).astype(np.uint8)
        vol = rng.integers(1, 20, n).astype(np.uint32)
        price = rng.integers(95, 106, n).astype(np.uint32)
        oid = np.zeros(n, np.uint64)
        if s >= 1:
            canc = rng.random(n) < 0.3
            action[canc] = 2
            oid[canc] = rng.integers(0, s * N // 2, int(canc.sum()))
        if s >= 3:  # from the fourth step on: modifications of earlier ids (a third price only, a third volume only, a third both)
            mod = (rng.random(n) < 0.1) & (action == 1)
            action[mod] = MOD
            side[mod] = rng.choice([2, 4, 6], int(mod.sum())).astype(np.uint8)
            oid[mod] = rng.integers(0, s * N // 2, int(mod.sum()))
        ins = (action, side, vol, np.zeros(n, np.uint32), price, oid)
        ids = env.submit_instructions_all(off, ins)
        for b, r in enumerate(refs):
            want = r.submit_instructions_native(tuple(a[b * N:(b + 1) * N] for a in ins))
            assert np.array_equal(ids[b * N:(b + 1) * N], want), (s, b)
        env.step()
        for r in refs:
            r.step()
        keyed_after.append(int(env.event_steps_keyed().sum()))
    assert not env.flags().any()
    h = env.history()
    for b, r in enumerate(refs):
        assert np.array_equal(h[:, b], r.history()), b
        got, want = env.trades(b, first=0), r.book.trades_array()
        assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), b
        got, want = env.orders(b), r.book.orders_array()
        assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), b
    per_step = np.diff([0] + keyed_after)
    assert list(per_step[:3]) == [B, B, B], per_step          # clean steps: keyed on the kernel without the modification code
    assert per_step[3] in (0, B) or per_step[3] < B, per_step   # the first step with modifications may still run on it: event by event
    assert list(per_step[4:]) == [B, B], per_step              # ... the hint has arrived (env.step() waits): keyed, modifications included
    env.close()


@pytest.mark.parametrize("assets,pool", [(2, 64), (3, 128)])
def test_markets_with_queues_longer_than_the_pool_and_modifications(bk, oracle, assets, pool):
    """The last combination of k_step_events' forms: a MARKET's joint queue (market_env.rs:110-121) longer than a book's pool -
    chunked - with cancellations and modifications across its assets: every book against the oracle's MarketEnv."""
    NM, T = 40, 5
    ticks = [1, 2, 5][:assets]
    env = bk.ManyMarketEnv(NM, 91, 0, ticks, 100_000, levels=10, max_live_orders=pool, max_orders=4096, trade_capacity=8192, history_capacity=T,
                           strict=False)
    ref = oracle.ManyMarkets(NM, 91, 0, ticks, 100_000, True, 10)
    rng = np.random.default_rng(assets * 7)
    for s in range(T):
        for m in range(NM):
            for _ in range(int(rng.integers(pool + 10, 2 * pool + 30))):
                a = int(rng.integers(0, assets))
                made = ref.book(m, a).n_orders()
                u = rng.random()
                if u < 0.38 and made:
                    oid = int(made - 1 - rng.integers(0, min(made, 50)))
                    env.cancel_order(m, a, oid)
                    ref.cancel_order(m, a, oid)
                elif u < 0.46 and made:
                    oid = int(made - 1 - rng.integers(0, min(made, 50)))
                    np_ = int(rng.integers(97, 104)) * ticks[a] if rng.random() < 0.5 else None
                    nv = int(rng.integers(1, 9)) if (np_ is None or rng.random() < 0.5) else None
                    env.modify_order(m, a, oid, np_, nv)
                    ref.modify_order(m, a, oid, np_, nv)
                else:
                    bid, vol = bool(rng.integers(0, 2)), int(rng.integers(1, 9))
                    price = None if rng.random() < 0.03 else int(rng.integers(98, 103)) * ticks[a]
                    assert env.place_order(m, a, bid, vol, 7, price) == ref.place_order(m, a, bid, vol, 7, price)
        env.step()
        ref.step()
    flags = env.flags().reshape(NM, assets)
    ok_m = ~flags.any(axis=1)  # (a market with a book whose pool overflowed dropped an order: not compared)
    assert ok_m.sum() >= 0.6 * NM, int(ok_m.sum())
    h, hr = env.history(), ref.history()
    for m in np.nonzero(ok_m)[0]:
        for a in range(assets):
            b = env.book(int(m), a)
            assert np.array_equal(h[:, b], hr[:, b]), (m, a)
            got, want = env.trades(b, first=0), ref.book(int(m), a).trades_array()
            assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), (m, a)
            got, want = env.orders(b), ref.book(int(m), a).orders_array()
            assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names), (m, a)
    keyed = env.event_steps_keyed().reshape(NM, assets)
    assert keyed[ok_m].sum() >= 0.5 * T * assets * int(ok_m.sum()), (int(keyed[ok_m].sum()), T * assets * int(ok_m.sum()))
    env.close()


@pytest.mark.parametrize("pool", [64, 128, 256])
def test_far_low_bids_keep_the_host_driven_step_on_the_keyed_loop(bk, oracle, pool):
    """Prices that span more than the key window (32 762 ticks): an external agent's stink bids far below the book.  The host-driven
    step takes the top-anchored window with those bids saturated (book_device.hpp keys_begin_wide: exact as long as no aggressor
    can reach them - a guard on the step's ask volume) instead of the event-by-event loop; a step whose asks COULD reach them, an ask
    below the window, or a modification in such a step falls back.  Every book against its oracle env."""
    far = [("place_order", True, 3 + i, 9, 1 + i) for i in range(5)]                      # bids at 1..5
    near = [("place_order", i % 2 == 0, 2 + i % 5, i, 50_000 + (i % 6) - (3 if i % 2 == 0 else 0)) for i in range(14)]
    busy = lambda s: [("place_order", (i + s) % 2 == 0, 1 + (i + s) % 4, i, 50_000 + ((i + s) % 5) - (2 if (i + s) % 2 == 0 else 0)) for i in range(10)] + [("cancel_order", 6 + s)]  # noqa: E731
    cases = {
        # (the first step of every book falls back: the guard counts the bid volume that RESTS when the step begins - none yet -
        # against the asks of the step that could take bids)
        "stink bids, ordinary flow": ([far + near, busy(1), busy(2) + [("place_order", False, 2, 3, None)], busy(3)], 3),
        "a market ask larger than the in-window bids reaches them": ([far + near, busy(1), [("place_order", False, 500, 3, None)], busy(3)], 2),
        "an ask below the window": ([far + near, busy(1) + [("place_order", False, 1, 3, 10_000)], busy(2)], 1),
        "a modification in such a step": ([far + near, busy(1) + [("modify_order", 7, None, 1)], busy(2)], 1),
        "cancelling the stink bids brings the narrow window back": ([far + near, [("cancel_order", i) for i in range(5)] + busy(1), busy(2)], 2),
    }
    names = list(cases)
    B, T = len(names), 4
    env = bk.ManyBookEnv(B, 77, 0, 1, 100_000, levels=10, max_live_orders=pool, max_orders=400, trade_capacity=800, history_capacity=T)
    refs = [oracle.StepEnv(77 + b, 0, 1, 100_000) for b in range(B)]
    for s in range(T):
        for b, name in enumerate(names):
            steps = cases[name][0]
            for f, *args in (steps[s] if s < len(steps) else []):
                getattr(env, f)(b, *args)
                getattr(refs[b], f)(*args)
        env.step()
        for r in refs:
            r.step()
    assert not env.flags().any(), dict(zip(names, env.flags().tolist()))
    _same_as_oracle(env, refs)
    keyed = dict(zip(names, env.event_steps_keyed().tolist()))
    for name in names:
        assert keyed[name] == cases[name][1], keyed
    env.close()


@pytest.mark.parametrize("pool", [64, 128, 256])
def test_far_high_asks_keep_the_host_driven_step_on_the_keyed_loop(bk, oracle, pool):
    """The mirror window (book_device.hpp keys_begin_wide_high): asks far ABOVE the book are saturated at the top price field; exact as
    long as no bid can reach them (a guard on the step's bid volume).  A market bid larger than the in-window asks, a bid above the
    window, a modification, or far orders on BOTH sides fall back.  Every book against its oracle env."""
    far = [("place_order", False, 3 + i, 9, 3_000_000 + 7 * i) for i in range(5)]        # asks ~60x above the book
    near = [("place_order", i % 2 == 0, 2 + i % 5, i, 50_000 + (i % 6) - (3 if i % 2 == 0 else 0)) for i in range(14)]
    busy = lambda s: [("place_order", (i + s) % 2 == 0, 1 + (i + s) % 4, i, 50_000 + ((i + s) % 5) - (2 if (i + s) % 2 == 0 else 0)) for i in range(10)] + [("cancel_order", 6 + s)]  # noqa: E731
    cases = {
        # (the first step of every book falls back: no ask volume rests yet when its guard is evaluated)
        "far asks, ordinary flow": ([far + near, busy(1), busy(2) + [("place_order", True, 2, 3, None)], busy(3)], 3),
        "a market bid larger than the in-window asks reaches them": ([far + near, busy(1), [("place_order", True, 500, 3, None)], busy(3)], 2),
        "a bid above the window": ([far + near, busy(1) + [("place_order", True, 1, 3, 2_000_000)], busy(2)], 1),
        "a modification in such a step": ([far + near, busy(1) + [("modify_order", 8, None, 1)], busy(2)], 1),
        "far orders on both sides": ([far + near + [("place_order", True, 4, 9, 5)], busy(1), busy(2)], 0),
        "cancelling the far asks brings the narrow window back": ([far + near, [("cancel_order", i) for i in range(5)] + busy(1), busy(2)], 2),
    }
    names = list(cases)
    B, T = len(names), 4
    env = bk.ManyBookEnv(B, 78, 0, 1, 100_000, levels=10, max_live_orders=pool, max_orders=400, trade_capacity=800, history_capacity=T)
    refs = [oracle.StepEnv(78 + b, 0, 1, 100_000) for b in range(B)]
    for s in range(T):
        for b, name in enumerate(names):
            steps = cases[name][0]
            for f, *args in (steps[s] if s < len(steps) else []):
                getattr(env, f)(b, *args)
                getattr(refs[b], f)(*args)
        env.step()
        for r in refs:
            r.step()
    assert not env.flags().any(), dict(zip(names, env.flags().tolist()))
    _same_as_oracle(env, refs)
    keyed = dict(zip(names, env.event_steps_keyed().tolist()))
    for name in names:
        assert keyed[name] == cases[name][1], keyed
    env.close()
