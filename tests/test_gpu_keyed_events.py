"""The host-driven step on the keyed loop (bourse_amd/csrc/step_events.hpp step_events_keyed, round 5).

`Env::step` over submitted instructions (ref crates/step_sim/src/env.rs:116-135; place / cancel / modify:
crates/order_book/src/orderbook.rs:583-611, 622-644, 743-772) runs on the slot-addressed assembly loops whenever a step has
no modification, fits one event per pool slot, and its prices fit the key window; everything else runs the event-by-event
loop.  Both must be the reference's step, and a book must be able to alternate between them: EVERY book of a batch is compared
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


def _drive(bk, oracle, pool, n_max, B, T, seed, p_market, p_mod, p_zero, tick=1, lo=94, hi=107, vols=None):
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
                    new_v = int(rng.integers(1, 40)) if rng.random() < 0.6 else None
                    env.modify_order(b, oid, new_p, new_v)
                    refs[b].modify_order(oid, new_p, new_v)
                    ok, n_ev = False, n_ev + 1
                else:
                    bid, trader = bool(rng.integers(0, 2)), int(rng.integers(0, 50))
                    vol = 0 if rng.random() < p_zero else (int(rng.integers(1, 40)) if vols is None else int(rng.choice(vols)))
                    price = None if rng.random() < p_market else int(rng.integers(lo, hi)) * tick
                    assert env.place_order(b, bid, vol, trader, price) == made[b] == refs[b].place_order(bid, vol, trader, price)
                    made[b] += 1
                    ok, n_ev = ok and vol != 0 and (price is not None or pool > 128), n_ev + 1
            busy[s, b], clean[s, b] = n_ev > 0, ok and 0 < n_ev <= pool
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
    env, refs, busy, clean = _drive(bk, oracle, pool, n_max, B, T, 500 + pool, p_market, p_mod=0.004, p_zero=0.002)
    _same_as_oracle(env, refs)
    keyed = env.event_steps_keyed()
    # every keyed step was a step the keyed form may take; and it took (nearly) all of those - what the calls cannot tell is
    # a pool without a spare slot, a resting order of volume 0 and the key window
    assert np.all(keyed <= clean.sum(axis=0)), "a step with a modification / volume 0 / too many events ran keyed"
    assert keyed.sum() >= 0.9 * clean.sum(), (int(keyed.sum()), int(clean.sum()))
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
        "modify": ([_std(), _std(1) + [("modify_order", 1, 101, None)], _std(2)], {2}),
        # (the order of volume 0 rests - orderbook.rs:430 never enters the match loop with it - and while it does, steps stay
        # on the event-by-event loop; whether step 2 finds it still there depends on the shuffle)
        "zero volume": ([_std(), _std(1) + [("place_order", True, 0, 3, 100)], _std(2)], {1, 2}),
        "more events than slots": ([_std(), [("cancel_order", 0)] * 70 + _std(1, 2), _std(2)], {2}),
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
    nothing; a modification for one asset sends only THAT asset's book to the event-by-event loop."""
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
                elif u < 0.36 and made:
                    oid = int(rng.integers(0, made))
                    nv = int(rng.integers(1, 30))
                    env.modify_order(m, a, oid, None, nv)
                    ref.modify_order(m, a, oid, None, nv)
                    ok[a] = False
                else:
                    bid, vol = bool(rng.integers(0, 2)), int(rng.integers(1, 30))
                    price = None if (rng.random() < 0.04 and pool > 128) else int(rng.integers(95, 106)) * ticks[a]
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
