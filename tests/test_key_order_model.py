"""Invariants of the order keys of the keyed event loop, on the integer model tools/key_model.py (the specification
bourse_amd/csrc/event_asm.hpp and book_device.hpp keys_begin follow): the best key IS the order the reference's
price-time priority picks, the prefix compare IS the inclusive crossing test, keys never collide - and they do without
the side bit (the bug the 8-bit test build of the library found)."""
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


def test_best_key_is_price_time_priority_and_prefix_compare_is_the_crossing_test():
    rng = random.Random(7)
    for _ in range(3000):
        seq_ctr = rng.randrange(1000, 2**31)
        lo = rng.choice([1, 50, 2**31, 2**32 - 40_000])
        span = rng.choice([1, 2, 7, 40, K.PSPAN + 1])
        orders = _book(rng, rng.randrange(0, 40), lo, span, seq_ctr, rng.choice([5, 300, K.SMASK - 200]))
        new_price, new_bid = lo + rng.randrange(span), rng.random() < 0.5
        prices = [o[0] for o in orders] + [new_price]
        assert K.window_ok(prices, [o[1] for o in orders], seq_ctr, 128)
        pbase, sbase = K.bases(prices, [o[1] for o in orders], seq_ctr)
        keys = [K.key(p, s, b, pbase, sbase) for p, s, b in orders]
        assert len(set(keys)) == len(keys) and all(k < K.MARKET_BID and k > K.MARKET_ASK for k in keys)
        asks = [(o, k) for o, k in zip(orders, keys) if not o[2]]
        bids = [(o, k) for o, k in zip(orders, keys) if o[2]]
        if asks:  # best ask: lowest price, then oldest (side.rs:300-313)
            assert min(asks, key=lambda x: x[1])[0] == min((o for o, _ in asks), key=lambda o: (o[0], o[1]))
        if bids:  # best bid: highest price, then oldest
            assert max(bids, key=lambda x: x[1])[0] == min((o for o, _ in bids), key=lambda o: (-o[0], o[1]))
        for k, (p, s, b) in zip(keys, orders):
            assert K.seq_of(k, b, sbase) == s
        # inclusive crossing test (orderbook.rs:430 / :463) in key space, the empty side included
        kp = K.prefix(new_price, new_bid, pbase)
        assert kp not in keys
        opp = [(o, k) for o, k in zip(orders, keys) if o[2] != new_bid]
        best = (min(k for _, k in opp) if new_bid else max(k for _, k in opp)) if opp else (K.DEAD if new_bid else 0)
        want = bool(opp) and (min(o[0] for o, _ in opp) <= new_price if new_bid else max(o[0] for o, _ in opp) >= new_price)
        assert K.crosses(kp, best, new_bid) == want
        # a market order's prefix crosses whatever is there and nothing when the side is empty
        assert K.crosses(K.MARKET_BID if new_bid else K.MARKET_ASK, best, new_bid) == bool(opp)
        # the order rests: prefix ^ arrival field, newer than everything live
        rested = kp ^ ((seq_ctr - sbase) << 1)
        assert rested == K.key(new_price, seq_ctr, new_bid, pbase, sbase) and rested not in keys


def test_without_the_side_bit_a_bid_and_an_ask_at_one_price_collide():
    pbase, sbase = 99, 0
    a = K.key(100, 100, False, pbase, sbase, side_bit=False)
    b = K.key(100, K.SMASK - 100, True, pbase, sbase, side_bit=False)
    assert a == b  # complementary arrival fields: the collision behind the zero-volume trades / endless match loop
    assert K.key(100, 100, False, pbase, sbase) != K.key(100, K.SMASK - 100, True, pbase, sbase)


def test_window_limits():
    assert not K.window_ok([0, 5], [], 10, 4)                      # price 0 is the ask market sentinel
    assert not K.window_ok([5, 0xFFFFFFFF], [], 10, 4)             # u32::MAX the bid one
    assert K.window_ok([1, 1 + K.PSPAN], [], 10, 4) and not K.window_ok([1, 2 + K.PSPAN], [], 10, 4)
    assert K.window_ok([5], [10], 10 + K.SMASK - 2 - 5, 4) and not K.window_ok([5], [10], 10 + K.SMASK - 1 - 4, 4)


def test_skip_bounds_stay_valid_and_never_skip_a_crossing_order():
    """The two bounds of the keyed loops (alo <= best ask key, bhi >= best bid key; event_asm.hpp EK_ALO / EK_BHI,
    book_device.hpp KeyState): exact after a reduction, valid after removals, pulled in at a rest.  Whatever the
    sequence of cancels, fills and new orders, a new order is only ever skipped when it really cannot cross."""
    rng = random.Random(11)
    for _ in range(300):
        pbase, sbase, seq = 99, 0, 1
        asks, bids = {}, {}          # key -> volume
        alo, bhi = 0, K.DEAD         # loosest
        for _step in range(200):
            op = rng.random()
            if op < 0.35 and (asks or bids):          # a cancellation / a fill: removals never touch the bounds
                side = asks if (asks and (not bids or rng.random() < 0.5)) else bids
                side.pop(rng.choice(list(side)))
            else:                                      # a new limit order
                is_bid = rng.random() < 0.5
                kp = K.prefix(100 + rng.randrange(12), is_bid, pbase)
                opp = asks if is_bid else bids
                truly = bool(opp) and (min(opp) <= kp if is_bid else max(opp) >= kp)
                skipped = kp < alo if is_bid else kp > bhi
                assert not (skipped and truly)
                vol = 1 + rng.randrange(3)
                if not skipped:
                    while vol and opp:                 # match loop: every reduction makes the bound exact
                        best = min(opp) if is_bid else max(opp)
                        if is_bid:
                            alo = best
                        else:
                            bhi = best
                        if not K.crosses(kp, best, is_bid):
                            break
                        t = min(vol, opp[best])
                        vol -= t
                        opp[best] -= t
                        if opp[best] == 0:
                            del opp[best]
                    else:
                        if vol and not opp:            # the reduction over an empty side returns its neutral element
                            if is_bid:
                                alo = K.DEAD
                            else:
                                bhi = 0
                if vol:                                # rests: its side's bound covers it
                    k = kp ^ (seq << 1)
                    seq += 1
                    if is_bid:
                        bids[k] = vol
                        bhi = max(bhi, k)
                    else:
                        asks[k] = vol
                        alo = min(alo, k)
            assert alo <= (min(asks) if asks else K.DEAD) and bhi >= (max(bids) if bids else 0)
