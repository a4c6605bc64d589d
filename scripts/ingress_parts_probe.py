"""Would the host-driven kernels gain from PARTS (as the agent pipelines do)?  The 8 192-book ingress stream as ONE env on one
stream against the same books as P envs of 8 192 / P books on P streams (each env's k_ingest / k_step_events then overlap the
others': a probe with no library change).  GPU box:  python scripts/ingress_parts_probe.py [books]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bourse_amd as bk
BT, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 48, 30
POOL = 512 if BT > 8192 else 256


def run(P):
    B = BT // P
    streams = [torch.cuda.Stream() for _ in range(P)]
    envs, data = [], []
    g = torch.Generator(device="cuda").manual_seed(0)
    for p in range(P):
        e = bk.ManyBookEnv(B, 1 + p * B, 0, 1, 100_000, levels=16, max_live_orders=POOL, max_orders=N * (T + 8), trade_capacity=64 * (T + 8),
                           strict=False, history_capacity=0, stream=streams[p].cuda_stream)
        e.enable_device_ingress(N)
        envs.append(e)
        n = B * N
        off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * N
        bs = []
        for s in range(T + 3):
            canc = (torch.rand(n, device="cuda", generator=g) < 0.3) if s else torch.zeros(n, dtype=torch.bool, device="cuda")
            bs.append((torch.where(canc, 2, 1).to(torch.int32), torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.uint8),
                       torch.randint(1, 30, (n,), device="cuda", generator=g, dtype=torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"),
                       torch.randint(90, 111, (n,), device="cuda", generator=g, dtype=torch.int32),
                       (torch.rand(n, device="cuda", generator=g) * max(1, int(s * N * 0.6))).to(torch.int64) * canc))
        data.append((off, bs, torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty((B, 2), dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()

    def join():  # every stream waits for every other's work so far: what parts INSIDE one env would pay after each call
        evs = []
        for p in range(P):
            ev = torch.cuda.Event()
            ev.record(streams[p])
            evs.append(ev)
        for p in range(P):
            for q in range(P):
                if q != p:
                    streams[p].wait_event(evs[q])

    def steps(lo, hi):
        for s in range(lo, hi):
            for p in range(P):
                off, bs, ids, st = data[p]
                with torch.cuda.stream(streams[p]):
                    envs[p].submit_instructions_device(off, *bs[s], out_ids=ids, status=st)
                    if not JOIN:
                        envs[p].step(sync=False)
            if JOIN:
                if P > 1:
                    join()
                for p in range(P):
                    with torch.cuda.stream(streams[p]):
                        envs[p].step(sync=False)
                if P > 1:
                    join()
    steps(0, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps(3, T + 3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = np.unique(np.concatenate([e.flags() for e in envs]))
    print(f"{'joined after every call: ' if JOIN else ''}{P} env(s) x {B} books on {P} stream(s): {dt / T * 1e3:.3f} ms/step -> {BT * T / dt / 1e6:.1f} M book-steps/s, flags {fl}", flush=True)
    for e in envs:
        e.close()


for JOIN in (False, True):
    for P in ((1, 2, 4, 1, 2, 4) if not JOIN else (2, 2)):
        run(P)
