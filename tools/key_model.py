"""Model of the order keys of the keyed event loops (bourse_amd/csrc/event_asm.hpp "KEYED event loop", the generated
R = 4, 8 loop, book_device.hpp "SIGNED KEYS" / keys_begin / key_window): the specification the device code follows, in plain
integers, so that its invariants can be checked exhaustively on the CPU (tests/test_key_order_model.py).

    ask key = 1 << 31 | (price - pbase) << 16 | (seq - sbase)              negative as an i32
    bid key =           (price - pbase) << 16 | 0xFFFF - (seq - sbase)     positive
    every other pool lane (free, cancelled, filled, still pending): 0
    best ask = signed MIN over ALL lanes, best bid = signed MAX over ALL lanes (no masks: the other side and the zeros
    lose, and an empty side answers with a value of the wrong sign)
    pbase = lowest price - 2 (price field >= 2; 1 is the market ask's), sbase = oldest live stamp - 1
    a new order's compare value: bid kp = 1 << 31 | field << 16 | 0xFFFF (crosses iff best ask <= kp),
                                 ask kp = field << 16 (crosses iff best bid >= kp); it rests as kp ^ sq,
                                 sq = 1 << 31 | arrival; the event word carries kp >> 16.
"""
SB = 16
SMASK = (1 << SB) - 1
PSPAN = (1 << 15) - 6
ASK = 1 << 31
DEAD = 0
MARKET_BID, MARKET_ASK = 0xFFFFFFFF, 0x10000  # compare values of market orders (AgentSet members' lists)


def i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & ASK else x


def window_ok(prices, seqs_live, seq_ctr, n_ev):
    """key_window: prices of the live and the new limit orders, stamps of the live ones."""
    if not prices:
        return True
    pmin, pmax = min(prices), max(prices)
    age = max((seq_ctr - s for s in seqs_live), default=0)
    return pmin >= 2 and pmax != 0xFFFFFFFF and pmax - pmin <= PSPAN and age + n_ev < SMASK - 1


def bases(prices, seqs_live, seq_ctr):
    pbase = (min(prices) - 2) if prices else 0xFFFFFFFD
    age = max((seq_ctr - s for s in seqs_live), default=0)
    return pbase, seq_ctr - age - 1


def event_word_half(price, is_bid, pbase):
    """pk: what key_event_words puts into the upper half of a new order's event word."""
    f = price - pbase
    assert 2 <= f < 0x7FFF
    return (0x8000 | f) if is_bid else f


def prefix(price, is_bid, pbase):
    """kp: the loop's one scalar instruction on the event word (s_or 0xFFFF for a bid, s_and 0xFFFF0000 for an ask)."""
    pk = event_word_half(price, is_bid, pbase)
    return ((pk << 16) | 0xFFFF) if is_bid else (pk << 16)


def key(price, seq, is_bid, pbase, sbase):
    s = seq - sbase
    assert 1 <= s <= SMASK - 2
    f = price - pbase
    return ((f << 16) | (0xFFFF - s)) if is_bid else (ASK | (f << 16) | s)


def best_of(keys_all_lanes, is_bid_aggressor):
    """The reduction: over EVERY pool lane, zeros and the aggressor's own side included."""
    return min(keys_all_lanes, key=i32) if is_bid_aggressor else max(keys_all_lanes, key=i32)


def crosses(kp, best, is_bid_aggressor):
    """The aggressor with compare value kp against the reduction's result (signed compares)."""
    return i32(best) <= i32(kp) if is_bid_aggressor else i32(best) >= i32(kp)


def rest(kp, arrival):
    return (kp ^ (ASK | arrival)) & 0xFFFFFFFF


def seq_of(k, sbase):
    f = k & 0xFFFF
    return sbase + (f if k & ASK else 0xFFFF - f)


# ---- round 5: a step whose prices do not fit the window (book_device.hpp keys_begin_wide, members' lists) -----------------
def wide_pbase(pmax):
    """Window anchored at the TOP price: in-window prices are pbase + 2 .. pmax, fields 2 .. PSPAN."""
    return pmax - PSPAN if pmax > PSPAN else 0


def wide_field(price, pbase):
    """Price field with the bids below the window SATURATED at 1 (asks are never below it: the set-up refuses such a step)."""
    return 1 if price < pbase + 2 else price - pbase


def wide_key(price, seq, is_bid, pbase, sbase):
    s, f = seq - sbase, wide_field(price, pbase)
    return ((f << 16) | (0xFFFF - s)) if is_bid else (ASK | (f << 16) | s)


def wide_prefix(price, is_bid, pbase, market=False):
    if market:
        return MARKET_BID if is_bid else MARKET_ASK
    f = wide_field(price, pbase)
    return (((0x8000 | f) << 16) | 0xFFFF) if is_bid else (f << 16)


def wide_guard(live, new, cancelled, pbase):
    """live: {slot: (price, vol, is_bid)} resting now; new: [(price, vol, is_bid, market)] of this step; cancelled: slots with a
    cancellation in this step's list.  None = "not this path" (an ask below the window); else (volume the asks can take,
    volume of the in-window bids that outlast the step): the keyed loop is exact iff the first does not exceed the second."""
    lowp = pbase + 2
    if any(not b and p < lowp for p, _, b in live.values()) or any(not b and not m and p < lowp for p, _, b, m in new):
        return None
    pbid = max([p for p, _, b in live.values() if b] + [p for p, _, b, m in new if b and not m], default=0)
    a_ask = sum(v for p, v, b, m in new if not b and (m or p <= pbid))
    w_bid = sum(v for s, (p, v, b) in live.items() if b and p >= lowp and s not in cancelled)
    return a_ask, w_bid
