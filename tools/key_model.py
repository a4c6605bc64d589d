"""Model of the order keys of the keyed event loop (bourse_amd/csrc/event_asm.hpp "KEYED event loop",
book_device.hpp keys_begin / key_window): the specification the device code follows, in plain integers, so that its
invariants can be checked exhaustively on the CPU (tests/test_key_order_model.py).

    key = (price - pbase) << (SB + 1) | s << 1 | side      asks (side 0): s = seq - sbase
                                                           bids (side 1): s = ~(seq - sbase) & SMASK
    pbase = lowest price - 1, sbase = oldest live stamp - 1; a new order's prefix kp = key with s = 0 (ask) / SMASK (bid);
    it rests as kp ^ (arrival << 1).
"""
SB = 16
SMASK = (1 << SB) - 1
PSPAN = (1 << (31 - SB)) - 4
DEAD = 0xFFFFFFFF
MARKET_BID, MARKET_ASK = 0xFFFFFFFE, 1  # prefixes of market orders (AgentSet members' lists)


def window_ok(prices, seqs_live, seq_ctr, n_ev):
    """key_window: prices of the live and the new limit orders, stamps of the live ones."""
    if not prices:
        return True
    pmin, pmax = min(prices), max(prices)
    age = max((seq_ctr - s for s in seqs_live), default=0)
    return pmin != 0 and pmax != 0xFFFFFFFF and pmax - pmin <= PSPAN and age + n_ev < SMASK - 1


def bases(prices, seqs_live, seq_ctr):
    pbase = (min(prices) - 1) if prices else 0xFFFFFFFE
    age = max((seq_ctr - s for s in seqs_live), default=0)
    return pbase, seq_ctr - age - 1


def prefix(price, is_bid, pbase, side_bit=True):
    kp = ((price - pbase) << SB) | (SMASK if is_bid else 0)
    return ((kp << 1) | (1 if is_bid else 0)) if side_bit else kp


def key(price, seq, is_bid, pbase, sbase, side_bit=True):
    s = seq - sbase
    assert 1 <= s <= SMASK - 2
    return prefix(price, is_bid, pbase, side_bit) ^ ((s << 1) if side_bit else s)


def crosses(kp, best, is_bid_aggressor):
    """The aggressor with prefix kp against the best key of the opposite side (its neutral element when empty)."""
    return best <= kp if is_bid_aggressor else best >= kp


def seq_of(k, is_bid, sbase):
    return sbase + (((k >> 1) ^ (SMASK if is_bid else 0)) & SMASK)
