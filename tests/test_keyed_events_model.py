"""tools/keyed_events_model.py: the order log REBUILT from four per-order facts (what step_events.hpp step_events_keyed does on
the GPU) equals the log an event-by-event book writes as it goes (ref crates/order_book/src/orderbook.rs:583-611, 622-644,
843-870), on random multi-step streams with cancellations of orders placed in the same step - before and after their
placement -, repeated cancellations, cancellations of filled orders, market orders and orders filled on arrival."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import keyed_events_model as M  # noqa: E402


def _stream(rng, made, n):
    evs, new_here = [], []
    for _ in range(n):
        u = rng.random()
        pool = list(range(max(0, made - 40), made)) + new_here
        if u < 0.4 and pool:
            evs.append(("cancel", rng.choice(pool)))
            if rng.random() < 0.15:
                evs.append(("cancel", evs[-1][1]))
        else:
            oid = made + len(new_here)
            price = None if rng.random() < 0.06 else rng.randint(95, 105)
            evs.append(("new", oid, rng.random() < 0.5, price, rng.randint(1, 30)))
            new_here.append(oid)
    rng.shuffle(evs)  # Env::step shuffles the queue: a cancellation may now come BEFORE the placement of the order it names
    return evs, made + len(new_here)


def test_rebuilt_log_equals_the_log_written_event_by_event():
    rng = random.Random(5)
    kinds = set()
    for case in range(300):
        ref, keyed = M.Book(), M.Book()
        made = 0
        for s in range(rng.randint(1, 8)):
            evs, made = _stream(rng, made, rng.randint(0, 30))
            t0 = s * 1000
            ref.step(t0, evs)
            written, observed = M.keyed_step(keyed, t0, evs)
            assert keyed.trades == ref.trades, (case, s)
            assert {i: o[:3] for i, o in keyed.rest.items()} == {i: o[:3] for i, o in ref.rest.items()}, (case, s)
            assert keyed.log == ref.log, (case, s, [i for i in ref.log if ref.log[i] != keyed.log.get(i)][:3])
            assert observed == ref.log  # (the model's own bookkeeping, for the record: the same loop observed as it went)
            for e in written.values():
                kinds.add((e["status"], e["key_t"] != 0, e["end"] == e["arr"]))
    # every branch of the rebuild occurred: active, filled on arrival, rested then filled, rested then cancelled, market remainder
    assert {(M.ACTIVE, True, False), (M.FILLED, False, True), (M.FILLED, True, False), (M.CANCELLED, True, False),
            (M.CANCELLED, False, True)} <= kinds
