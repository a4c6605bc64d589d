"""Fuzz the device-resident instruction ingress over fresh seeds: random book counts, markets of 1-3 assets with random
tick sizes, ragged batches with new / cancel / modify / null instructions, bad prices planted in random books, queue
several submits per step - device entry (bk_submit_instructions_device) and, since round 5, the HOST-array entry of the
same flow (bk_submit_instructions_host as tickets, two in flight, results fetched one submit late) against the
per-order host entries (themselves fuzzed against the oracle by scripts/fuzz_host.py).  GPU box.  FUZZ_LO / FUZZ_HI."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bourse_amd as bk
import test_gpu_device_ingress as D

U64MAX = 2**64 - 1
bad = n = 0
lo, hi = int(os.environ.get("FUZZ_LO", 0)), int(os.environ.get("FUZZ_HI", 300))
for seed in range(lo, hi):
    rng = np.random.default_rng(770_000 + seed)
    M = int(rng.choice([1, 1, 2, 3]))
    NM = int(rng.integers(1, 120))
    ticks = [int(rng.choice([1, 2, 5])) for _ in range(M)]
    lcm = int(np.lcm.reduce(ticks))
    T, nmax, subs = int(rng.integers(1, 7)), int(rng.integers(1, 20)), int(rng.integers(1, 3))
    qcap = nmax * subs * M + int(rng.integers(0, 5))  # never full here (the capacity path: tests/test_gpu_device_ingress.py)
    # (512 slots: k_step_events<8> switches to its form with the keyed modifications when k_ingest's hint arrives - round 6)
    kw = dict(levels=int(rng.integers(1, 20)), max_live_orders=int(rng.choice([256, 512])), max_orders=nmax * subs * T * M + 8, trade_capacity=nmax * subs * T * 2 + 8,
              history_capacity=T)
    try:
        if M == 1:
            dev = bk.ManyBookEnv(NM, seed, 0, ticks[0], 100_000, stream=torch.cuda.current_stream().cuda_stream, strict=False, **kw)
            hst = bk.ManyBookEnv(NM, seed, 0, ticks[0], 100_000, strict=False, **kw)
            host = bk.ManyBookEnv(NM, seed, 0, ticks[0], 100_000, strict=False, **kw)
            hv = host
        else:
            dev = bk.ManyMarketEnv(NM, seed, 0, ticks, 100_000, stream=torch.cuda.current_stream().cuda_stream, strict=False, **kw)
            hst = bk.ManyMarketEnv(NM, seed, 0, ticks, 100_000, strict=False, **kw)
            host = bk.ManyMarketEnv(NM, seed, 0, ticks, 100_000, strict=False, **kw)
            hv = D._BookView(host)
        dev.enable_device_ingress(qcap)
        hst.enable_device_ingress(qcap)
        pend = None  # (ticket, expected ids / applied / codes) of the host-array env's previous submit

        def settle(p):
            ids, stt, _ = hst.submit_result(p[0])
            assert np.array_equal(stt[:, 0], p[3]) and np.array_equal(stt[:, 1], p[2]), (seed, "host-array status")
            assert np.array_equal(ids, p[1]), (seed, "host-array ids")
        B = NM * M
        counts = np.zeros(B, dtype=np.int64)
        for s in range(T):
            for _ in range(subs):
                bad_books = [int(b) for b in rng.integers(0, B, size=int(rng.integers(0, 3)))]
                off, book_of, ins = D._stream_step(rng, B, counts, nmax, lcm, bad_books)
                action = ins[0]
                out_ids = torch.full((len(action),), -1, dtype=torch.int64, device="cuda")
                status = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
                dev.submit_instructions_device(D._dev(torch, off.astype(np.int64)), D._dev(torch, action), *[D._dev(torch, x) for x in ins[1:]],
                                               out_ids=out_ids, status=status)
                tk = hst.submit_instructions_all_async(off, (action,) + tuple(ins[1:]))
                if pend is not None:
                    settle(pend)
                want_ids, want_applied, want_code = D._apply_host(hv, book_of, off, (action,) + tuple(ins[1:]), B)
                pend = (tk, want_ids, want_applied, want_code)
                st = status.cpu().numpy().view(np.uint32)
                assert np.array_equal(st[:, 0], want_code), (seed, s, "codes", st[:, 0], want_code)
                assert np.array_equal(st[:, 1], want_applied), (seed, s, "applied")
                assert np.array_equal(out_ids.cpu().numpy().view(np.uint64), want_ids), (seed, s, "ids")
                counts += np.bincount(book_of[want_ids != U64MAX], minlength=B)
            dev.step(sync=False)
            hst.step(sync=False)
            host.step()
        if pend is not None:
            settle(pend)
        dev.sync()
        hst.sync()
        assert np.array_equal(hst.flags(), host.flags()) and np.array_equal(hst.history(), host.history()), (seed, "host-array history")
        assert np.array_equal(hst.trade_counts(), host.trade_counts()), (seed, "host-array trade counts")
        assert np.array_equal(dev.flags(), host.flags()), (seed, "flags")
        assert np.array_equal(dev.history(), host.history()), (seed, "history")
        assert np.array_equal(dev.trade_counts(), host.trade_counts()), (seed, "trade counts")
        for b in sorted(set(int(x) for x in rng.integers(0, B, size=4))):
            gd, gh, ga = dev.trades(b, first=0), host.trades(b, first=0), hst.trades(b, first=0)
            od, oh, oa = dev.orders(b), host.orders(b), hst.orders(b)
            for f in gd.dtype.names:
                assert np.array_equal(gd[f], gh[f]) and np.array_equal(ga[f], gh[f]), (seed, b, f)
            for f in od.dtype.names:
                assert np.array_equal(od[f], oh[f]) and np.array_equal(oa[f], oh[f]), (seed, b, f)
        n += 1
        dev.close(); host.close(); hst.close()
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAIL", str(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "ERR", type(e).__name__, str(e)[:300], flush=True)
print("device ingress fuzz:", n, "configurations ok, failures:", bad)
