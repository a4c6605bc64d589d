"""Device-resident ingress for MARKETS (ManyMarketEnv: M assets per market, one shuffled queue per market): every step an agent layer
on the GPU hands over N instructions per book; the rate with the library as shipped, or - BOURSE_AMD_LIBRARY=<a -DBOURSE_AMD_EV_KEYED=0
build> - with every step on the event-by-event loop.  GPU box:  python scripts/market_ingress_rate.py [markets [assets]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bourse_amd as bk
NM, M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, T = 24, 24
B = NM * M
env = bk.ManyMarketEnv(NM, 1, 0, [1] * M, 100_000, levels=16, max_live_orders=256, max_orders=N * (T + 8), trade_capacity=64 * (T + 8), strict=False,
                       history_capacity=0, stream=torch.cuda.current_stream().cuda_stream)
env.enable_device_ingress(N * M)
g = torch.Generator(device="cuda").manual_seed(0)
off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * N
n = B * N


def make(s):
    canc = (torch.rand(n, device="cuda", generator=g) < 0.3) if s else torch.zeros(n, dtype=torch.bool, device="cuda")
    return (torch.where(canc, 2, 1).to(torch.int32), torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.uint8),
            torch.randint(1, 30, (n,), device="cuda", generator=g, dtype=torch.int32), torch.zeros(n, dtype=torch.int32, device="cuda"),
            torch.randint(95, 106, (n,), device="cuda", generator=g, dtype=torch.int32),
            (torch.rand(n, device="cuda", generator=g) * max(1, int(s * N * 0.6))).to(torch.int64) * canc)


batches = [make(s) for s in range(T + 3)]
ids, st = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty((B, 2), dtype=torch.int32, device="cuda")
for s in range(T + 3):
    if s == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    env.submit_instructions_device(off, *batches[s], out_ids=ids, status=st)
    env.step(sync=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = np.unique(env.flags())
assert int(st[:, 0].max()) == 0 and not fl.any(), fl
print(f"{NM} markets x {M} assets x {N} instructions per book-step: {dt / T * 1e3:.3f} ms/step -> {B * T / dt / 1e6:.1f} M book-steps/s, "
      f"{env.event_steps_keyed().sum() / (B * (T + 3)) * 100:.0f} % of the book-steps keyed, trades/book-step {env.trade_counts().sum() / (B * (T + 3)):.1f}")
