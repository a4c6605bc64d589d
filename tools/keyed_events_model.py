"""Plain-Python model of what bourse_amd/csrc/step_events.hpp step_events_keyed RE-DERIVES instead of observing: the order log
of a host-driven step (ref crates/order_book/src/orderbook.rs:583-611 place, 622-644 cancel, 429-487 match, 843-870
match_orders) from four per-order facts collected around a slot-addressed event loop, and the up-front resolution of a
cancellation's target.  tests/test_keyed_events_model.py checks both against a straightforward event-by-event book on random
streams (the device code itself is checked on the GPU: tests/test_gpu_keyed_events.py).

A step: events in shuffled order, each ("new", id, bid, price | None, vol) or ("cancel", id); volumes > 0 (the keyed form
refuses a step with a volume of 0 anywhere)."""
NEW, ACTIVE, FILLED, CANCELLED = 0, 1, 2, 3
NEVER = 1 << 62  # "no end time yet"


class Book:
    """The reference's book, event by event: price-time priority, a log entry per order."""

    def __init__(self):
        self.rest = {}   # id -> [bid, price, vol, seq]
        self.log = {}    # id -> dict(status, vol, price, arr, end, key_t)
        self.seq = 0
        self.trades = []  # (t, passive id, aggressor id, vol, price)

    def _best(self, bid_side):
        c = [(i, o) for i, o in self.rest.items() if o[0] == bid_side]
        if not c:
            return None
        return min(c, key=lambda io: ((-io[1][1]) if bid_side else io[1][1], io[1][3]))[0]

    def new(self, t, oid, bid, price, vol):
        market = price is None
        v = vol
        while v > 0:
            p = self._best(not bid)
            if p is None:
                break
            o = self.rest[p]
            if not market and (o[1] > price if bid else o[1] < price):
                break
            tv = min(v, o[2])
            o[2] -= tv
            v -= tv
            self.trades.append((t, p, oid, tv, o[1]))
            self.log[p]["vol"] = o[2]
            if o[2] == 0:
                self.log[p].update(status=FILLED, end=t)
                del self.rest[p]
        e = dict(status=ACTIVE, vol=v, price=price, arr=t, end=NEVER, key_t=0)
        if v == 0:
            e.update(status=FILLED, end=t)
        elif market:
            e.update(status=CANCELLED, end=t)  # the remainder of a market order is dropped (orderbook.rs:521-524)
        else:
            e["key_t"] = t
            self.rest[oid] = [bid, price, v, self.seq]
            self.seq += 1
        self.log[oid] = e

    def cancel(self, t, oid):
        if oid in self.rest:  # only an Active order changes (orderbook.rs:622-644)
            self.log[oid].update(status=CANCELLED, vol=self.rest[oid][2], end=t)
            del self.rest[oid]

    def step(self, t0, events):
        for k, ev in enumerate(events):
            if ev[0] == "new":
                self.new(t0 + k, *ev[1:])
            else:
                self.cancel(t0 + k, ev[1])


def keyed_step(book, t0, events):
    """The same step the way step_events_keyed runs it; returns the log entries it would write {id: fields}.

    Up front: every new order gets a "slot" (here: its id), every cancellation is resolved to the order it names if that order is
    live now or new in this step, else to nothing.  The loop then only sees slots: a cancellation clears whatever rests in the
    slot at that moment.  Afterwards the log is rebuilt from arrpos / first cancellation after arrival / last passive trade /
    left on arrival + the final volume and liveness."""
    live0 = set(book.rest)
    new_ids = {ev[1] for ev in events if ev[0] == "new"}
    arrpos, vol0, first_cancel, last_passive, left = {}, {}, {}, {}, {}
    for k, ev in enumerate(events):
        if ev[0] == "new":
            arrpos[ev[1]] = k
            vol0[ev[1]] = left[ev[1]] = ev[4]
    for k, ev in enumerate(events):
        if ev[0] == "cancel" and (ev[1] in live0 or ev[1] in new_ids) and k > arrpos.get(ev[1], -1):
            first_cancel[ev[1]] = min(first_cancel.get(ev[1], 1 << 30), k)
    n_tr = len(book.trades)
    # the slot-addressed loop: same matching, no log
    shadow_log = book.log
    book.log = {i: dict(e) for i, e in shadow_log.items()}  # (the model's Book writes a log as it goes: on a copy)
    for k, ev in enumerate(events):
        if ev[0] == "new":
            book.new(t0 + k, *ev[1:])
        elif ev[1] in live0 or ev[1] in new_ids:
            book.cancel(t0 + k, ev[1])  # clears the slot if something rests there NOW (a key that is still 0 otherwise)
    observed = book.log
    book.log = shadow_log
    for t, p, a, tv, _ in book.trades[n_tr:]:
        last_passive[p] = max(last_passive.get(p, -1), t - t0)
        left[a] -= tv
    out = {}
    for oid in sorted(live0 | new_ids):
        alive = oid in book.rest
        vol = book.rest[oid][2] if alive else None
        if oid in new_ids:
            price = next(ev[3] for ev in events if ev[0] == "new" and ev[1] == oid)
            t_arr = t0 + arrpos[oid]
            if price is None or left[oid] == 0:  # never rested
                e = dict(status=FILLED if left[oid] == 0 else CANCELLED, vol=left[oid], price=price, arr=t_arr, end=t_arr, key_t=0)
            else:
                e = dict(price=price, arr=t_arr, key_t=t_arr)
                if alive:
                    e.update(status=ACTIVE, vol=vol, end=NEVER)
                else:
                    # dead with volume left = cancelled; the device reads the volume from the slot's register, where a fill to 0
                    # leaves 0 and a cancellation leaves the remainder - the model asks the passive trades instead
                    filled = _filled_passively(book, n_tr, oid, left[oid])
                    e.update(status=FILLED if filled else CANCELLED, vol=0 if filled else _vol_at_cancel(book, n_tr, oid, left[oid]),
                             end=t0 + (last_passive[oid] if filled else first_cancel[oid]))
            out[oid] = e
        else:
            touched = oid in last_passive or not alive
            if not touched:
                continue
            e = dict(book.log[oid])
            if alive:
                e["vol"] = vol
            else:
                start = shadow_log[oid]["vol"]
                filled = _filled_passively(book, n_tr, oid, start)
                e.update(status=FILLED if filled else CANCELLED, vol=0 if filled else _vol_at_cancel(book, n_tr, oid, start),
                         end=t0 + (last_passive[oid] if filled else first_cancel[oid]))
            out[oid] = e
    for oid, e in out.items():
        shadow_log[oid] = e
    return out, observed


def _passive_total(book, n_tr, oid):
    return sum(tv for _, p, _, tv, _ in book.trades[n_tr:] if p == oid)


def _filled_passively(book, n_tr, oid, start):
    return _passive_total(book, n_tr, oid) == start


def _vol_at_cancel(book, n_tr, oid, start):
    return start - _passive_total(book, n_tr, oid)
