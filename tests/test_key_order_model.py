"""Invariants of the order keys of the keyed event loops, on the integer model tools/key_model.py (the specification
bourse_amd/csrc/event_asm.hpp, the generated R = 4, 8 loop and book_device.hpp keys_begin follow): the signed minimum /
maximum over ALL pool lanes - dead lanes (0) and the other side included - IS the order the reference's price-time
priority picks, the signed compare against the event word's value IS the inclusive crossing test (an empty side
included), keys never collide, and the skip bounds never skip an order that crosses."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import key_model as K


def _book(rng, n, lo, span, seq_ctr, max_age):
    orders = []
    seqs = rng.sample(range(seq_ctr - max_age, seq_ctr), min(n, max_age))
    for s in seqs:
        orders.append((lo + rng.randrange(span), s, rng.random() < 0.5))
    return orders


def test_best_key_is_price_time_priority_and_the_signed_compare_is_the_crossing_test():
    rng = random.Random(7)
    for _ in range(3000):
        seq_ctr = rng.randrange(1000, 2**31)
        lo = rng.choice([2, 50, 2**31, 2**32 - 40_000])
        span = rng.choice([1, 2, 7, 40, K.PSPAN + 1])
        orders = _book(rng, rng.randrange(0, 40), lo, span, seq_ctr, rng.choice([5, 300, K.SMASK - 200]))
        new_price, new_bid = lo + rng.randrange(span), rng.random() < 0.5
        prices = [o[0] for o in orders] + [new_price]
        assert K.window_ok(prices, [o[1] for o in orders], seq_ctr, 128)
        pbase, sbase = K.bases(prices, [o[1] for o in orders], seq_ctr)
        keys = [K.key(p, s, b, pbase, sbase) for p, s, b in orders]
        assert len(set(keys)) == len(keys) and K.DEAD not in keys
        assert all((K.i32(k) < 0) == (not o[2]) for k, o in zip(keys, orders))  # asks negative, bids positive
        lanes = keys + [K.DEAD] * 3  # the reduction runs over every pool lane: dead ones, pending ones (0), both sides
        asks = [(o, k) for o, k in zip(orders, keys) if not o[2]]
        bids = [(o, k) for o, k in zip(orders, keys) if o[2]]
        if asks:  # best ask: lowest price, then oldest (side.rs:300-313)
            want = min((o for o, _ in asks), key=lambda o: (o[0], o[1]))
            assert orders[keys.index(K.best_of(lanes, True))] == want
        if bids:  # best bid: highest price, then oldest
            want = min((o for o, _ in bids), key=lambda o: (-o[0], o[1]))
            assert orders[keys.index(K.best_of(lanes, False))] == want
        for k, (p, s, b) in zip(keys, orders):
            assert K.seq_of(k, sbase) == s
        # inclusive crossing test (orderbook.rs:430 / :463) in key space, the empty side included
        kp = K.prefix(new_price, new_bid, pbase)
        assert kp not in keys and kp >> 16 == K.event_word_half(new_price, new_bid, pbase)
        opp = [o for o in orders if o[2] != new_bid]
        best = K.best_of(lanes, new_bid)
        want = bool(opp) and (min(o[0] for o in opp) <= new_price if new_bid else max(o[0] for o in opp) >= new_price)
        assert K.crosses(kp, best, new_bid) == want
        # a market order's compare value crosses whatever is there and nothing when the side is empty; no limit order has it
        assert K.crosses(K.MARKET_BID if new_bid else K.MARKET_ASK, best, new_bid) == bool(opp)
        assert kp != (K.MARKET_BID if new_bid else K.MARKET_ASK)
        # the order rests: compare value ^ (sign | arrival field), newer than everything live
        rested = K.rest(kp, seq_ctr - sbase)
        assert rested == K.key(new_price, seq_ctr, new_bid, pbase, sbase) and rested not in keys


def test_a_bid_and_an_ask_at_one_price_never_collide():
    # (round 2's first keys had no side bit: complementary arrival fields collided - zero-volume trades, an endless match
    # loop; the sign separates the sides here whatever the fields are)
    pbase, sbase = 98, 0
    for s in (1, 100, K.SMASK - 100, K.SMASK - 2):
        for t in (1, 100, K.SMASK - 100, K.SMASK - 2):
            assert K.key(100, s, False, pbase, sbase) != K.key(100, t, True, pbase, sbase)


def test_window_limits():
    assert not K.window_ok([0, 5], [], 10, 4)                      # price 0 is the ask market sentinel
    assert not K.window_ok([1, 5], [], 10, 4)                      # (pbase = lowest - 2 must not wrap)
    assert not K.window_ok([5, 0xFFFFFFFF], [], 10, 4)             # u32::MAX the bid one
    assert K.window_ok([2, 2 + K.PSPAN], [], 10, 4) and not K.window_ok([2, 3 + K.PSPAN], [], 10, 4)
    assert K.window_ok([5], [10], 10 + K.SMASK - 2 - 5, 4) and not K.window_ok([5], [10], 10 + K.SMASK - 1 - 4, 4)
    # the widest window still leaves the market bid's field (0x7FFF) to itself
    assert K.event_word_half(2 + K.PSPAN, True, 0) < 0xFFFF


def test_skip_bounds_stay_valid_and_never_skip_a_crossing_order():
    """The two bounds of the keyed loops (alo <= best ask key, bhi >= best bid key as i32; event_asm.hpp EK_ALO / EK_BHI,
    book_device.hpp KeyState): exact after a reduction, valid after removals, pulled in at a rest.  Whatever the
    sequence of cancels, fills and new orders, a new order is only ever skipped when it really cannot cross."""
    rng = random.Random(11)
    for _ in range(300):
        pbase, sbase, seq = 98, 0, 1
        asks, bids = {}, {}          # key -> volume
        alo, bhi = -(1 << 31), (1 << 31) - 1   # loosest
        for _step in range(200):
            op = rng.random()
            if op < 0.35 and (asks or bids):          # a cancellation / a fill: removals never touch the bounds
                side = asks if (asks and (not bids or rng.random() < 0.5)) else bids
                side.pop(rng.choice(list(side)))
            else:                                      # a new limit order
                is_bid = rng.random() < 0.5
                kp = K.prefix(100 + rng.randrange(12), is_bid, pbase)
                opp = asks if is_bid else bids
                lanes = list(asks) + list(bids) + [K.DEAD]
                best = K.best_of(lanes, is_bid)
                truly = bool(opp) and K.crosses(kp, best, is_bid)
                skipped = K.i32(kp) < alo if is_bid else K.i32(kp) > bhi
                assert not (skipped and truly)
                vol = 1 + rng.randrange(3)
                if not skipped:
                    while vol:                         # match loop: every reduction makes the bound exact
                        best = K.best_of(list(asks) + list(bids) + [K.DEAD], is_bid)
                        if is_bid:
                            alo = K.i32(best)
                        else:
                            bhi = K.i32(best)
                        if not K.crosses(kp, best, is_bid):
                            break
                        assert best in opp             # a cross is always against a live order of the other side
                        t = min(vol, opp[best])
                        vol -= t
                        opp[best] -= t
                        if opp[best] == 0:
                            del opp[best]
                if vol:                                # rests: its side's bound covers it
                    k = K.rest(kp, seq)
                    seq += 1
                    if is_bid:
                        bids[k] = vol
                        bhi = max(bhi, K.i32(k))
                    else:
                        asks[k] = vol
                        alo = min(alo, K.i32(k))
            assert alo <= (min(map(K.i32, asks)) if asks else 0) and bhi >= (max(map(K.i32, bids)) if bids else 0)


def test_the_model_and_the_device_header_agree_on_the_constants():
    """tools/key_model.py is the specification; book_device.hpp / event_asm.hpp / the generator carry the same numbers."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "bourse_amd", "csrc", "book_device.hpp")).read()
    m = re.search(r"KP_MKT_BID = (0x[0-9A-Fa-f]+)u, KP_MKT_ASK = (0x[0-9A-Fa-f]+)u", hdr)
    assert m and (int(m.group(1), 16), int(m.group(2), 16)) == (K.MARKET_BID, K.MARKET_ASK)
    assert re.search(r"KEY_ASK = 0x80000000u", hdr) and K.ASK == 0x80000000
    # price window: (1 << 15) - 6 at the shipped 16-bit arrival field; arrival window 0xFFFF
    m = re.search(r"KEY_PSPAN = \(1u << \(31u - \(KEY_SB > 16u \? KEY_SB : 16u\)\)\) - (\d+)u", hdr)
    assert m and (1 << 15) - int(m.group(1)) == K.PSPAN
    assert "pbase = pmin - 2u" in hdr and "pmin >= 2u" in hdr
    # the loops: a bid's compare value is ew | 0xFFFF, an ask's ew & 0xFFFF0000; both rest as kp ^ sq; a market ask's is 0x10000
    asm = open(os.path.join(root, "bourse_amd", "csrc", "event_asm.hpp")).read()
    gen = open(os.path.join(root, "tools", "gen_event_asm.py")).read()
    for text in (asm, gen):
        assert "0xffff0000" in text and ", 0xffff" in text and "s_xor_b32" in text
    assert '"-1" if agg_bid else "0x10000"' in gen


def test_saturated_low_bids_never_change_a_pick_when_the_liquidity_guard_holds():
    """Round 5 (keys_begin_wide): a window anchored at the top price with the bids below it saturated at price field 1, plus
    the up-front guard, picks exactly what price-time priority on the TRUE prices picks - over random steps of new limit /
    market orders and cancellations in shuffled order on books that hold far-low bids (MomentumAgent bids clamped to 0)."""
    import random

    rng = random.Random(5)
    exact = refused = 0
    for trial in range(1500):
        mid = rng.randrange(40_000, 60_000)
        seq_ctr, live = 100, {}
        for s in range(rng.randrange(4, 40)):  # resting book: near orders around the mid, some bids far below
            is_bid = rng.random() < 0.6
            far = is_bid and rng.random() < 0.35
            price = rng.choice([0, 1, 7, mid - 40_000]) if far else (mid - rng.randrange(1, 200) if is_bid else mid + rng.randrange(1, 200))
            live[s] = [price, rng.randrange(1, 120), is_bid, seq_ctr]
            seq_ctr += 1
        new = []
        for _ in range(rng.randrange(1, 25)):
            is_bid, market = rng.random() < 0.5, rng.random() < 0.3
            off = rng.randrange(-60, 200)
            price = (0xFFFFFFFF if is_bid else 0) if market else (mid - off if is_bid else mid + off)
            if not market and is_bid and rng.random() < 0.2:
                price = rng.choice([0, 3, mid - 39_000])
            new.append((price, rng.randrange(1, 150), is_bid, market))
        cancelled = set(rng.sample(sorted(live), rng.randrange(0, max(1, len(live) // 3))))
        prices = [p for p, _, _, _ in live.values()] + [p for p, _, _, m in new if not m]
        pbase = K.wide_pbase(max(prices))
        g = K.wide_guard({s: (p, v, b) for s, (p, v, b, _) in live.items()}, new, cancelled, pbase)
        if g is None or g[0] > g[1]:
            refused += 1
            continue
        exact += 1
        sbase = min(q for _, _, _, q in live.values()) - 1
        events = [("new", i) for i in range(len(new))] + [("cancel", s) for s in cancelled]
        rng.shuffle(events)
        ref = {s: list(o) for s, o in live.items()}   # the reference: true prices
        dev = {s: list(o) for s, o in live.items()}   # the keyed loop: keys with saturation
        nxt, sq = max(live) + 1, seq_ctr
        for kind, i in events:
            if kind == "cancel":
                ref.pop(i, None), dev.pop(i, None)
                continue
            price, vol, is_bid, market = new[i]
            v_ref = v_dev = vol
            kp = K.wide_prefix(price, is_bid, pbase, market)
            while v_ref > 0:  # reference pick: best price, then oldest, on the other side
                cand = [(s, o) for s, o in ref.items() if o[2] != is_bid]
                if not cand:
                    break
                s_ref, o = min(cand, key=lambda so: ((so[1][0] if is_bid else -so[1][0]), so[1][3]))
                if (price < o[0]) if is_bid else (price > o[0]):
                    break
                keys = [K.wide_key(o2[0], o2[3], o2[2], pbase, sbase) for o2 in dev.values()] + [K.DEAD]
                best = K.best_of(keys, is_bid)
                assert v_dev > 0 and K.crosses(kp, best, is_bid), (trial, "the keyed loop stops where the reference trades")
                s_dev = next(s for s, o2 in dev.items() if K.wide_key(o2[0], o2[3], o2[2], pbase, sbase) == best)
                assert s_dev == s_ref, (trial, "different passive order", s_dev, s_ref)
                assert o[0] >= pbase + 2 or not o[2], (trial, "an aggressor reached a saturated bid")
                tv = min(v_ref, o[1])
                for book, s in ((ref, s_ref), (dev, s_dev)):
                    book[s][1] -= tv
                    if book[s][1] == 0:
                        del book[s]
                v_ref -= tv
                v_dev -= tv
            else:
                pass
            if v_dev > 0:  # the keyed loop stops too where the reference stopped
                keys = [K.wide_key(o2[0], o2[3], o2[2], pbase, sbase) for o2 in dev.values()] + [K.DEAD]
                assert not K.crosses(kp, K.best_of(keys, is_bid), is_bid) or not any(o2[2] != is_bid for o2 in dev.values()), trial
            if v_ref > 0 and not market:
                ref[nxt] = [price, v_ref, is_bid, sq]
                dev[nxt] = [price, v_dev, is_bid, sq]
                nxt, sq = nxt + 1, sq + 1
        assert ref == dev, trial
    assert exact > 500 and refused > 50, (exact, refused)
