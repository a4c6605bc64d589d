# What k_step_events spends on its phases: variant builds that leave a phase out (-DBOURSE_AMD_EV_SKIP=bits; results are then
# wrong, only the kernel time counts): 1 the shuffle's swaps, 2 the order-log writes, 4 matching, 7 all three.  GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "" 1 2 4 7; do
  if [ -n "$v" ]; then export BOURSE_AMD_LIBRARY=$R/build_variants/lib_evskip$v.so; else unset BOURSE_AMD_LIBRARY; fi
  echo -n "skip=${v:-0}: "; python3 - <<PY 2>/dev/null | tail -1
import sys, os, numpy as np, torch
sys.path.insert(0, "$R")
import bourse_amd as bk
B, N, T = 8192, 48, 12
env = bk.ManyBookEnv(B, 1, 0, 1, 100_000, levels=16, max_live_orders=256, max_orders=N * (T + 8), trade_capacity=64 * (T + 8), strict=False, history_capacity=0, stream=torch.cuda.current_stream().cuda_stream)
env.enable_device_ingress(N)
g = torch.Generator(device="cuda").manual_seed(0); n = B * N
off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * N
def make(s):
    canc = (torch.rand(n, device="cuda", generator=g) < 0.3) if s else torch.zeros(n, dtype=torch.bool, device="cuda")
    return (torch.where(canc, 2, 1).to(torch.int32), torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.uint8), torch.randint(1, 30, (n,), device="cuda", generator=g, dtype=torch.int32),
            torch.zeros(n, dtype=torch.int32, device="cuda"), torch.randint(90, 111, (n,), device="cuda", generator=g, dtype=torch.int32), (torch.rand(n, device="cuda", generator=g) * max(1, int(s * N * 0.6))).to(torch.int64) * canc)
bs = [make(s) for s in range(T)]
for s in range(4):
    env.submit_instructions_device(off, *bs[s]); env.step(sync=False)
env.sync(); env.profile(1)
for s in range(4, T):
    env.submit_instructions_device(off, *bs[s]); env.step(sync=False)
env.sync(); ms, nl = env.profile_read_kind(3)
print("k_step_events %.3f ms per launch (%d launches)" % (ms / nl, nl))
PY
done
