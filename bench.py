#!/usr/bin/env python3
"""Headline benchmark: book-steps/sec of the many-book LOB step simulator on MI355X.

Workload (BASELINE.json configs[2], SURVEY §8d "C3"): 65 536 independent books per GPU x 128
on-device RandomAgents (64 x rate 0.8 vol [10,20) + 64 x rate 0.2 vol [50,70), ticks [32,64)),
32 levels/side, tick 2, step_size 100 000, seed 101 + global book index.  One "step" = one
agents.update + Env::step for EVERY book (ref crates/step_sim/src/runner.rs:58-59).

  python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

N > 1: one rank per GPU over RCCL.  Started plainly (`python bench.py --gpus N`), the parent process - before it
imports torch or touches a GPU - starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
and exits with its code; started by torch.distributed.run itself (RANK / WORLD_SIZE in the environment) it is a rank.
Books are independent, so each rank steps its own contiguous shard with NO data-path collective; the only exchange is a
64-byte market-stats all-gather per launch.  Default = BASELINE configs[3]: STRONG scaling, 65 536 books in
total, 65 536 / N per GPU, seeds by global book index; `--scaling weak` keeps 65 536 books per GPU.

Timing: W warm-up steps, barrier + synchronize, K timed steps, synchronize (each rank's time; the job's = the MAX over
ranks), barrier.  Right before the warm-up the env's public `bk_warm` entry runs scratch steps in chunks (`--preheat-steps`
per chunk, grown to >= 10 ms of load; 0 = off) until three consecutive chunk rates agree within 1 % and at least
`--preheat-min-ms` (200) have passed (capped; what ran is
reported as the top-level `preheat_steps` and `config.preheat_chunk_rates`): the GPU's clocks fall within milliseconds
of idling and need tens of ms of load to come back - how many differs from box to box - which a short timed region (the
driver's `--steps 20 --warmup 5` = 6 ms) would otherwise measure.  bk_warm steps the env's own books with its own
kernels and puts everything back (state, level-2 records, step counter; no history slot, no trade record): the
simulated steps are exactly the W + K ones.

Prints ONE JSON line (rank 0) incl. `roofline` (HIP-event kernel time vs. HBM peak) and, at
N = 1, `cpu_baseline` (the CPU oracle = literal restatement of the reference algorithm, timed on
this machine's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
N_CU, CLOCK_GHZ = 256, 2.4  # same guide: 256 CUs, 2.4 GHz peak engine clock

WORKLOADS = {
    # name: (books_per_gpu, levels, groups)                                   SURVEY §8d
    "C2": (4096, 16, [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)]),
    "C3": (65536, 32, [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)]),
    "C5": (8192, 64, [(256, (100, 164), (10, 20), 2, 0.8), (256, (100, 164), (50, 70), 2, 0.2)]),
}
# BASELINE configs[4] as written: momentum + "market-maker" agents (the reference's only liquidity provider is NoiseAgent;
# it has no market-maker type), doc-example parameters (ref crates/step_sim/src/lib.rs:53-73) scaled to 512 agents
MOM_P = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=20.0, scale=0.5, order_ratio=1.0,
             price_dist_mu=0.0, price_dist_sigma=10.0)
NOISE_P = dict(tick_size=2, p_limit=0.3, p_market=0.2, p_cancel=0.2, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
WORKLOADS["C5M"] = (8192, 64, [("momentum", 0, 256, MOM_P), ("noise", 256, 256, NOISE_P)])
TICK, STEP_SIZE, SEED = 2, 100_000, 101
# External agents: every step an agent layer ON THE GPU hands over 48 instructions per book (70 %
# new limit orders around 100, 30 % cancellations of earlier ids) as the six SoA arrays of `submit_instructions`
# (rust/src/step_sim_numpy.rs:233-275), bk_submit_instructions_device queues them, bk_step_async is Env::step.  The flow
# lets the books fill up, so a run is at most 33 steps (scripts/device_ingress_rate.py is the same stream).
WORKLOADS["INGRESS"] = (8192, 16, None)
INGRESS_N, INGRESS_MAX_STEPS = 48, 33


def algorithmic_bytes(kind, S, W4, ev, tr, new, spl=1, pipe="split", n_ins=0.0):
    """COMPULSORY HBM bytes per book-step of one step kernel -> (hbm_bytes, l2_bytes).

    hbm_bytes is what `roofline.achieved` is computed from: only what a launch MUST read from and write to HBM in the layout
    shipped (S = the per-book state block, W4 = the level-2 record, 32 B per trade record, 2 B per event word, 8 B per new
    order of a step batch).  l2_bytes are the working bytes a kernel round-trips through global memory but that stay
    cache-resident at the sizes they occur (the wave-parallel decode's 1.3 KB lane-state record and its ~1 KB block-start
    spills): rounds 2 - 5 charged them as HBM bytes, and PMC showed two kernels "moving" more algorithmic bytes than the
    counters saw (C2 k_run_wave 3.5 KB claimed vs 1.0 KB measured).  tests/test_roofline_accounting.py
    asserts hbm_bytes <= 1.05 x the FETCH_SIZE / WRITE_SIZE traffic of every kernel in profiles/pmc_traffic.json.
    ev / tr / new: events, trades and new orders per book-step (measured by the run); spl: steps per launch of a fused kernel;
    n_ins: instructions per book-step (k_ingest)."""
    rec = 2.0 * 1280.0  # lane-state record of the wave-parallel decode, in and out
    if kind in ("k_run_random", "k_run_mixed", "k_run_wave"):
        # fused: the book block in and out once per launch, the L2 record and the trade records every step
        hbm = 2.0 * S / spl + W4 + 32.0 * tr
        return hbm, ((rec / spl + 2.5 * 1024.0) if (kind == "k_run_wave" or pipe == "wave") else 0.0)
    if kind == "k_agents_fsm":  # 32 B in (RNG + live masks); RNG + N_ev + the shuffled list + the new orders out
        return 32.0 + 20.0 + 2.0 * ev + 8.0 * new, 0.0
    if kind == "k_agents_wave":  # header line in; RNG state + step batch (256 B header + list + new orders) out
        return 256.0 + 16.0 + 256.0 + 2.0 * ev + 8.0 * new, rec + 2.5 * 1024.0
    if kind == "k_agents_mixed_lanes":
        # RNG + live-mask line + touches in; the members' lists in and out (2-byte slots, about one entry per resting order),
        # 16 B per new order into the pool, the shuffled event list out
        return 192.0 + 4.0 * ev + 16.0 * new + 2.0 * ev, 0.0
    if kind == "k_agents_mixed_wave":  # header line in; the members' lists in and out, new orders into the pool, the batch out
        return 256.0 + 256.0 + 4.0 * ev + 16.0 * new + 256.0 + 2.0 * ev, rec + 6.0 * 1024.0
    if kind == "k_step_batch":  # the state block in and out, the batch in, the L2 record and the trade records out
        return 2.0 * S + 64.0 + 2.0 * ev + 8.0 * new + W4 + 32.0 * tr, 0.0
    if kind == "k_step_events":
        # state in and out, the last and the new L2 record, 20 B per queued event, trade records, and the ORDER LOG the path
        # keeps (orderbook.rs:113-115: every order's status / volume / times / key): the ten words of a new order's entry, two
        # words (volume + status or end time) per cancellation and per fill of a resting order.  Round 6's toggled PMC passes
        # (profiles/r06/pmc_step_events.txt): without the log the kernel moves 0.99 x the rest of this sum, the log adds 2.1 KB
        # of writes at 48 events per book-step - rounds 4 - 5 left it out of the sum and read it as 1.36 x "wasted" traffic.
        # l2: the wave-parallel shuffle's lane-state record (1.0 KB read per step, written back when the block changed)
        return 2.0 * S + 2.0 * W4 + 20.0 * ev + 32.0 * tr + 40.0 * new + 8.0 * max(ev - new, 0.0) + 8.0 * tr, 1280.0 + 320.0
    if kind == "k_ingest":  # the six arrays in, event records + the new orders' immutable halves and log entries out
        return 27.0 * n_ins + 16.0 * ev + 80.0 * new, 0.0
    raise KeyError(kind)


def ingress_batch(torch, g, n, N, s, f_mod=0.0, f_mkt=0.0):
    """Step s of the external-agents stream (--workload INGRESS): n elements = N per book, as the six SoA arrays of
    `submit_instructions` (rust/src/step_sim_numpy.rs:233-275) on the device.  Per element: 30 % cancellations of earlier ids (from
    the second step on), then f_mod modifications of earlier ids (Env::modify_order, orderbook.rs:743-772: a third price only, a
    third volume only, a third both - side bit 1 = has price, bit 2 = has volume), f_mkt market orders (price u32::MAX for a bid / 0
    for an ask, orderbook.rs:594-606), the rest new limit orders at 90..110.  With both fractions 0 the draws are round 5's stream
    exactly.  (tests/test_gpu_device_ingress.py steps this very stream against one oracle env per book.)"""
    ACT_MODIFY = 0x80000003 - (1 << 32)  # BK_ACTION_MODIFY as the int32 the 4-byte array holds
    u = torch.rand(n, device="cuda", generator=g) if (s or f_mod > 0.0 or f_mkt > 0.0) else torch.ones(n, device="cuda")
    canc = (u < 0.3) if s else torch.zeros(n, dtype=torch.bool, device="cuda")
    ids = (torch.rand(n, device="cuda", generator=g) * max(1, int(s * N * 0.6))).to(torch.int64)
    side = torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.uint8)
    vol = torch.randint(1, 30, (n,), device="cuda", generator=g, dtype=torch.int32)
    price = torch.randint(90, 111, (n,), device="cuda", generator=g, dtype=torch.int32)
    action = torch.where(canc, 2, 1).to(torch.int32)
    if f_mod > 0.0 or f_mkt > 0.0:
        mod = (u >= 0.3) & (u < 0.3 + f_mod) if s else torch.zeros(n, dtype=torch.bool, device="cuda")
        mkt = (u >= 0.3 + f_mod) & (u < 0.3 + f_mod + f_mkt)
        which = torch.randint(0, 3, (n,), device="cuda", generator=g, dtype=torch.uint8)
        mside = torch.where(which == 0, 2, torch.where(which == 1, 4, 6)).to(torch.uint8)
        action = torch.where(mod, ACT_MODIFY, action).to(torch.int32)
        side = torch.where(mod, mside, side)
        price = torch.where(mkt, torch.where(side == 1, -1, 0).to(torch.int32), price)
        ids = ids * (canc | mod)
    else:
        ids = ids * canc
    return action, side, vol, torch.zeros(n, dtype=torch.int32, device="cuda"), price, ids


def bench_ingress(args, torch):
    import bourse_amd as bk

    B = args.books or WORKLOADS["INGRESS"][0]
    N, LV = INGRESS_N, WORKLOADS["INGRESS"][1]
    W, K = args.warmup, args.steps
    if W + K > INGRESS_MAX_STEPS:
        raise SystemExit(f"--workload INGRESS: warm-up + steps <= {INGRESS_MAX_STEPS} (the flow fills the pools; e.g. --steps 24 --warmup 6)")
    pool = 512 if B > 8192 else 256
    n = B * N
    T = W + K

    def new_env():
        e = bk.ManyBookEnv(B, 1, 0, 1, STEP_SIZE, levels=LV, max_live_orders=pool, max_orders=N * (T + 8), trade_capacity=64 * (T + 8),
                           strict=False, history_capacity=0, stream=torch.cuda.current_stream().cuda_stream)
        e.enable_device_ingress(N)
        return e

    g = torch.Generator(device="cuda").manual_seed(0)
    off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * N

    f_mod, f_mkt = float(args.modify_frac), float(args.market_frac)

    def make(s):
        return ingress_batch(torch, g, n, N, s, f_mod, f_mkt)

    batches = [make(s) for s in range(T)]  # generated ahead: the rate is the library's, not torch's RNG
    out_ids = torch.empty(n, dtype=torch.int64, device="cuda")
    status = torch.empty((B, 2), dtype=torch.int32, device="cuda")

    def run(env, lo, hi):
        for s in range(lo, hi):
            env.submit_instructions_device(off, *batches[s], out_ids=out_ids, status=status)
            env.step(sync=False)

    # clocks: the same scratch-step pre-heat as the agent workloads needs an agent set; here a throw-away env runs the stream
    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < args.preheat_min_ms and args.preheat_steps > 0:
        e = new_env()
        run(e, 0, T)
        torch.cuda.synchronize()
        e.close()
    env = new_env()
    run(env, 0, W)
    torch.cuda.synchronize()
    keyed0 = float(env.event_steps_keyed().sum())  # (the keyed fraction is the TIMED region's: on 512-slot pools the library switches
    torch.cuda.synchronize()                       # to the kernel with the keyed modifications a step or more after the first one)
    t0 = time.perf_counter()
    run(env, W, T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(status[:, 0].max()) != 0 or env.flags().any():
        raise SystemExit(f"INGRESS: status / capacity flags set ({np.unique(env.flags())}): the rate would be of a run that dropped orders")
    keyed = (float(env.event_steps_keyed().sum()) - keyed0) / (B * K)
    tr_total = int(env.trade_counts().sum())
    env.close()
    # kernel times: the last steps of the same stream on a fresh env, every launch between two events on the env's stream
    P = min(12, K)
    env = new_env()
    run(env, 0, T - P)
    torch.cuda.synchronize()
    tc0 = int(env.trade_counts().sum())
    env.profile(1)
    evs = []
    for s in range(T - P, T):  # (no synchronize between the steps: a launch that starts on an idle GPU reads ~10 us long)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env.submit_instructions_device(off, *batches[s], out_ids=out_ids, status=status)
        b.record()
        env.step(sync=False)
        evs.append((a, b))
    torch.cuda.synchronize()
    ing_ms = sum(a.elapsed_time(b) for a, b in evs)
    ev_ms, nl = env.profile_read_kind(3)
    env.profile_read()
    tr = (int(env.trade_counts().sum()) - tc0) / (B * P)
    new = sum(int((batches[s][0] == 1).sum()) for s in range(T - P, T)) / (B * P)
    S, W4 = env.state_bytes_per_book(), env.width * 4
    env.close()
    acct = {"S": S, "W4": W4, "ev": float(N), "tr": tr, "new": new, "spl": 1, "pipe": "ingress", "n_ins": float(N)}
    ev_bytes, _ = algorithmic_bytes("k_step_events", **acct)         # per book-step (DESIGN.md 2's table)
    ing_bytes, _ = algorithmic_bytes("k_ingest", **acct)             # arrays in, event records + new orders' records out
    ev_launch_ms, ing_launch_ms = ev_ms / max(nl, 1), ing_ms / P
    ach = ev_bytes * B / (ev_launch_ms * 1e-3) / 1e9
    traffic = traffic_src = None  # (PMC passes of this command, committed and replayed: as for the agent workloads)
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pkey = f"INGRESS{'MIX' if (f_mod, f_mkt) == (0.05, 0.02) else ''}/{B}"  # (the mixed stream's record is of THESE fractions)
        rec = allp.get(pkey, {}).get("k_step_events", {}) if (f_mod, f_mkt) in ((0.0, 0.0), (0.05, 0.02)) else {}
        if "hbm_bytes_per_book_step" in rec:
            traffic = rec["hbm_bytes_per_book_step"] * B
            traffic_src = f"profiles/pmc_traffic.json[{pkey}]: {allp.get('_source', '')}; replayed, not measured in this run"
    except Exception:  # noqa: BLE001
        pass
    line = {
        "metric": "book-steps/sec", "value": B * K / dt, "unit": "book-steps/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic", "keyed_frac": keyed,
        "config": {"workload": f"INGRESS: {B} books x {N} instructions per book-step (30 % cancellations of earlier ids, "
                               f"{100 * f_mod:g} % modifications of earlier ids, {100 * f_mkt:g} % market orders, the rest new limit orders at "
                               f"90..110) as device arrays through bk_submit_instructions_device + bk_step_async, {pool}-slot pools, "
                               f"{LV} levels" + ("; stream = scripts/device_ingress_rate.py" if f_mod == f_mkt == 0.0 else ""),
                   "modify_frac": f_mod, "market_frac": f_mkt,
                   "books_total": B, "instructions_per_book_step": N, "instructions_per_s": n * K / dt,
                   "trades_per_book_step": tr_total / (B * T), "keyed_step_fraction": keyed,
                   "pipeline": "k_ingest + k_step_events per step (host-driven kernels; no agents pipeline)",
                   "preheat": f"the whole {T}-step stream on throw-away envs for >= {args.preheat_min_ms:.0f} ms"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": traffic_src, "kernel": "k_step_events", "avg_launch_ms": ev_launch_ms, "launches": int(nl), "bytes_per_book_step": ev_bytes,
                     "book_steps_per_launch": B, "accounting": acct,
                     "launches_sampled_in": f"the last {P} steps of the same stream on a fresh env, HIP events around every launch",
                     # aggregate: both kernels' compulsory bytes of a step / the step's wall time in the timed region
                     "achieved_node": (ev_bytes + ing_bytes) * B / (dt / K) / 1e9, "peak_node": HBM_PEAK_GBPS,
                     "frac_node": (ev_bytes + ing_bytes) * B / (dt / K) / 1e9 / HBM_PEAK_GBPS,
                     "kernels": {"k_step_events": {"avg_launch_ms": ev_launch_ms, "bytes_per_book_step": ev_bytes,
                                                   "frac": ach / HBM_PEAK_GBPS},
                                 "k_ingest": {"avg_launch_ms": ing_launch_ms, "bytes_per_book_step": ing_bytes,
                                              "frac": ing_bytes * B / (ing_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}},
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = _cpu_baseline_ingress(batches, B, N, T)
    print(json.dumps(line))


def _cpu_baseline_ingress(batches, B, N, T, sample=256):
    """The same instruction stream through the CPU oracle (one StepEnv per book: place / cancel calls + step) for the first
    `sample` books, one thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    nb = min(sample, B)
    dts = (np.uint32, np.uint8, np.uint32, np.uint32, np.uint32, np.uint64)
    host = [[[np.ascontiguousarray(x[b * N:(b + 1) * N].cpu().numpy(), dtype=dt) for x, dt in zip(step, dts)] for b in range(nb)] for step in batches]
    envs = [pyoracle.StepEnvNumpy(1 + b, 0, 1, STEP_SIZE) for b in range(nb)]
    t = time.perf_counter()
    for s in range(T):
        for b, e in enumerate(envs):
            e.submit_instructions_native(host[s][b])  # (the loop over the arrays runs in the library, as the reference's in Rust)
            e.step()
    d = time.perf_counter() - t
    return {"value": nb * T / d, "unit": "book-steps/s", "cores": 1, "kind": "port",
            "sample": f"the first {nb} books x {T} steps of the same instruction stream: one oracle StepEnvNumpy per book, one native "
                      f"submit_instructions + one step per book-step (oracle/libbourse_oracle.so -O3 through ctypes), one thread"}


def effective_cores():
    """Host threads this process can really run at once: the scheduler affinity capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs but cap the container at 16 CPUs' worth of time - 256 threads then only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2: "<quota|max> <period>"
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    used = max(1, min(n, int(quota))) if quota else n
    return used, n, quota


def cpu_baseline(groups, levels, n_books, budget_s=12.0):
    if any(isinstance(g[0], str) for g in groups):
        return _cpu_baseline(dict(members=groups), levels, n_books, budget_s)
    return _cpu_baseline(dict(groups=groups), levels, n_books, budget_s)


def _cpu_baseline(agents_kw, levels, n_books, budget_s=12.0):
    """Time the CPU oracle (kind "port": C++ restatement of the reference algorithm, ordered maps per
    side, one Env per book) on all host cores, on a bounded sample of the same workload: the workload's own number of
    books (so that the working set is the real one, not a cache-resident toy), a few steps, three timed repetitions.
    Books are constructed by the threads that step them (per-thread allocator arenas, first-touch placement).  For
    RandomAgents workloads the batched SoA CPU implementation (oracle/bourse_soa.cpp: ladder + per-level FIFO, checked
    equal to the oracle by tests/test_soa_cpu.py) is timed beside it under "soa": the oracle measures the reference's
    ALGORITHM, the SoA figure what the same host does with the kernel's data structure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    cores, visible, quota = effective_cores()
    probe = pyoracle.ManyBooks(64, SEED, 0, TICK, STEP_SIZE, True, levels, **agents_kw)
    t = time.perf_counter()
    probe.run(30, 1)
    rate1 = 64 * 30 / (time.perf_counter() - t)
    books = int(min(n_books, 65536))
    est = rate1 * min(cores, books) * 0.5
    warm = 10
    steps = int(max(5, min(100, est * budget_s / 3.0 / books - warm / 3.0)))
    many = pyoracle.ManyBooks(books, SEED, 0, TICK, STEP_SIZE, True, levels, build_threads=cores, **agents_kw)
    many.run(warm, cores)
    vals = []
    for _ in range(3):
        t = time.perf_counter()
        many.run(steps, cores)
        vals.append(books * steps / (time.perf_counter() - t))
    del many
    out = {
        "value": float(np.median(vals)), "unit": "book-steps/s", "cores": cores, "kind": "port",
        "sample": f"{books} books x {steps} steps, median of 3 repetitions after {warm} warm-up steps, same agents/levels/"
                  f"seeds, {cores} host threads (books statically partitioned, built by their own threads), "
                  f"oracle/libbourse_oracle.so -O3",
        "values": vals, "single_thread_probe": rate1,
        "thread_efficiency": float(np.median(vals)) / (rate1 * cores),
        "host": f"{visible} logical CPUs visible, cgroup CPU quota {quota if quota else 'none'}: {cores} threads used",
    }
    if "groups" in agents_kw:
        try:
            out["soa"] = _cpu_baseline_soa(pyoracle, agents_kw["groups"], levels, books, cores)
        except Exception as e:  # the SoA engine covers bounded-grid RandomAgents shapes only
            out["soa"] = {"error": str(e)}
    return out


def _cpu_baseline_soa(pyoracle, groups, levels, books, cores, budget_s=6.0):
    steps1 = 20
    p1 = pyoracle.SoaBooks(256, SEED, 0, TICK, STEP_SIZE, levels, groups, history_capacity=steps1, trade_reserve=0, threads=1)
    p1.run(steps1)          # touches the trade / history memory
    p1.clear_trades()
    t = time.perf_counter()
    p1.run(steps1)
    rate1 = 256 * steps1 / (time.perf_counter() - t)
    steps = int(max(5, min(50, rate1 * min(cores, books) * 0.3 * budget_s / 4.0 / books)))
    soa = pyoracle.SoaBooks(books, SEED, 0, TICK, STEP_SIZE, levels, groups, history_capacity=steps, trade_reserve=0,
                            threads=cores)
    soa.run(steps)          # warm-up of the same length: every page the timed passes write is resident
    vals = []
    for _ in range(3):
        soa.clear_trades()  # the consumer has drained the records (as the GPU bench does per launch)
        t = time.perf_counter()
        soa.run(steps)
        vals.append(books * steps / (time.perf_counter() - t))
    return {
        "value": float(np.median(vals)), "unit": "book-steps/s", "cores": cores, "kind": "soa",
        "sample": f"{books} books x {steps} steps, median of 3 repetitions after {steps} warm-up steps, L2 record of every "
                  f"step and all trade records kept; {cores} host threads, oracle/libbourse_soa.so -O3",
        "values": vals, "single_thread_probe": rate1, "thread_efficiency": float(np.median(vals)) / (rate1 * cores),
    }


def spawn_ranks(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` started by hand: run the N ranks as a child torch.distributed.run job.  Called before
    torch is imported - a process that has initialised the GPU must never exec or fork into another program."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def rank_watchdog(seconds: float):
    """A rank that hangs - RCCL's rendezvous or communicator init, a collective a peer never joins, a stream probe - must END,
    non-zero, so that torch.distributed.run tears the job down and the parent propagates the failure (the
    first real multi-GPU run is also the first test of all of these).  faulthandler's timer thread prints every thread's stack
    and calls _exit(1) when `seconds` pass; `rank_watchdog(0)` disarms it."""
    import faulthandler

    if seconds > 0:
        faulthandler.dump_traceback_later(seconds, exit=True)
    else:
        faulthandler.cancel_dump_traceback_later()


def init_ranks(dist, torch, backend, local_rank, timeout_s):
    """init_process_group with a bounded timeout.  RCCL: first with `device_id` (the communicator is created HERE, so a rank
    whose device or xGMI link is unusable fails at a named place instead of in the first collective); if that raises, once
    more without it (lazy communicator on the current device) before giving up."""
    import datetime

    to = datetime.timedelta(seconds=timeout_s)
    if backend != "nccl":
        dist.init_process_group(backend=backend, timeout=to)
        return "gloo"
    try:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=to)
        return "nccl (communicator created at init on cuda:%d)" % local_rank
    except Exception as e:  # noqa: BLE001
        print(f"[bench rank {os.environ.get('RANK')}] init_process_group(nccl, device_id=cuda:{local_rank}) failed: {e!r}; "
              f"retrying without device_id", file=sys.stderr, flush=True)
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        dist.init_process_group(backend="nccl", timeout=to)
        return "nccl (lazy communicator)"


def selftest_gloo(args, books_total_default):
    """CPU-only check of the multi-rank plumbing (tests/test_bench_spawn.py): rendezvous, shard arithmetic and the
    64-byte stats all-gather over gloo, with NO stepping (there is no CPU execution path to step with)."""
    import torch
    import torch.distributed as dist

    from bourse_amd import parallel

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank_watchdog(args.rank_timeout)
    # (tests/test_bench_spawn.py: one rank fails before / hangs inside the rendezvous - the job must end non-zero, soon)
    fault = os.environ.get("BOURSE_AMD_BENCH_TEST_FAULT", "")
    if fault == f"exit:{rank}":
        raise SystemExit(f"rank {rank}: injected failure before init_process_group")
    if fault == f"hang:{rank}":
        time.sleep(10_000)
    init_ranks(dist, torch, "gloo", 0, args.dist_timeout)
    total = args.books or books_total_default
    first, B = parallel.shard_books(total, rank, world) if args.scaling == "strong" else (rank * total, total)
    rec = parallel.pack_stats({"n_books": B, "sum_trade_vol": 0, "sum_trades": first, "sum_events": 0, "sum_bid_vol": 0,
                               "sum_ask_vol": 0, "min_bid": 0xFFFFFFFF, "max_bid": 0, "min_ask": 0xFFFFFFFF, "max_ask": 0})
    out = parallel.all_gather_records(torch.from_numpy(rec.copy()), dist)
    g = parallel.combine_stats(out.numpy())
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "gloo", "n_gpus": world, "scaling": args.scaling, "books_total": g["n_books"],
                          "ranks_seen": int(out.shape[0]), "first_books_sum": g["sum_trades"]}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--books", type=int, default=0,
                    help="books: in TOTAL with --scaling strong, per GPU with --scaling weak (default: the workload's)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's books sharded over the GPUs (BASELINE configs[3]); weak = that many per GPU")
    ap.add_argument("--selftest-gloo", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-ranks", action="store_true",
                    help="N > 1 on a box with ONE GPU: every rank steps its shard on GPU 0 and the collectives run over gloo. "
                         "Executes the whole multi-rank code path except RCCL-over-xGMI (rendezvous, N concurrent stream probes, "
                         "sharding, region / barrier logic, stats all-gather, consistency checks); the rate it prints is that of "
                         "N processes sharing one GPU, NOT a measurement of N GPUs (the line says `dry_ranks: true`)")
    ap.add_argument("--steps-per-launch", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--modify-frac", type=float, default=0.0,
                    help="--workload INGRESS: fraction of the instructions that are modifications of earlier ids (Env::modify_order)")
    ap.add_argument("--market-frac", type=float, default=0.0, help="--workload INGRESS: fraction that are market orders")
    ap.add_argument("--l1-gather", action="store_true", help="also all-gather every book's L1 record per launch (SURVEY 8e ii)")
    ap.add_argument("--no-history", action="store_true", help="keep only the latest L2 record (diagnostic)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "fused", "split", "wave_split", "wave"])
    ap.add_argument("--wave-parts", type=int, default=0, help="parts the wave pipeline cuts the batch in (0 = library default)")
    ap.add_argument("--profile-every", type=int, default=8, help="HIP-event-time every Nth step's kernels (0 = none)")
    ap.add_argument("--preheat-steps", type=int, default=50,
                    help="scratch steps per CHUNK of the library's bk_warm entry right before the warm-up (state restored "
                         "afterwards), so that the timed region starts at steady clocks; chunks repeat until three consecutive "
                         "chunk rates agree within 1 %% (0 = no pre-heat)")
    ap.add_argument("--preheat-max-chunks", type=int, default=40, help="cap of the adaptive pre-heat (1 = one fixed chunk)")
    ap.add_argument("--preheat-min-ms", type=float, default=200.0,
                    help="the first pre-heat also lasts at least this long: on some boxes the chunk rates agree within 1 %% "
                         "after 40 ms while the regions behind still climb 5 %% for another ~60 ms (profiles/r05/bench_ramp_probe.txt)")
    ap.add_argument("--repeats", type=int, default=4, help="extra timed regions of --steps after the reported one (median in `runs`)")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="N > 1: seconds a rendezvous / collective may take before it raises")
    ap.add_argument("--rank-timeout", type=float, default=900.0,
                    help="N > 1: seconds after which a rank that has not finished dumps its stacks and exits non-zero (0 = never)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))  # nothing has touched torch or the GPU yet
    if args.selftest_gloo:
        return selftest_gloo(args, WORKLOADS[args.workload][0])

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (bourse_amd has no CPU path)")
    if args.dry_ranks:
        local_rank = 0  # every rank on the one GPU there is
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: exercises RCCL path)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rank_watchdog(args.rank_timeout)
        # (RCCL refuses two ranks on one device: in a dry run the 64-byte records travel over gloo instead)
        dist_backend = init_ranks(dist, torch, "gloo" if args.dry_ranks else "nccl", local_rank, args.dist_timeout)
    if args.workload == "INGRESS":
        if world > 1:
            raise SystemExit("--workload INGRESS is a one-GPU line (books shard as for the agent workloads: bourse_amd/parallel.py)")
        return bench_ingress(args, torch)
    coll_dev = "cpu" if args.dry_ranks else "cuda"  # where the small bookkeeping tensors of the collectives live

    def all_ranks(x):
        """Every rank's value of a python float (one all-gather); [x] without a process group."""
        if dist is None:
            return [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        out = torch.empty(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(out, t)
        return [float(v) for v in out.cpu()]

    import bourse_amd
    from bourse_amd import parallel

    books_default, levels, groups = WORKLOADS[args.workload]
    if args.scaling == "strong":  # the workload's books in total, contiguous shards (8 x 8 192)
        books_total = args.books or books_default
        first_book, B = parallel.shard_books(books_total, rank, world)
        if B == 0:
            raise SystemExit("fewer books than GPUs")
    else:
        B = args.books or books_default
        first_book, books_total = rank * B, world * B
    mixed = any(isinstance(g[0], str) for g in groups)
    n_agents = sum(g[2] if isinstance(g[0], str) and g[0] != "random" else (g[1] if isinstance(g[0], str) else g[0])
                   for g in groups)
    spl = max(1, min(args.steps_per_launch, args.steps))
    hist_cap = 0 if args.no_history else spl
    trade_cap = max(64, n_agents // 2 * 3 // 2) * spl  # ~35 trades/book-step measured at C3 (128 agents); overflow is flagged and checked below
    stream = torch.cuda.current_stream().cuda_stream
    env = bourse_amd.ManyBookEnv(B, SEED, 0, TICK, STEP_SIZE, True, levels=levels, max_live_orders=min(n_agents, 512),
                                 trade_capacity=trade_cap, history_capacity=hist_cap,
                                 book_offset=first_book, device=local_rank, stream=stream, strict=False)
    if mixed:
        env.set_agents(groups)
    else:
        env.set_random_agents(groups)
    env.set_pipeline(args.pipeline)
    if args.wave_parts:
        env.set_wave_options(64, args.wave_parts)
    pipe, parts = env.pipeline()
    gather = parallel.StatsGather(env, dist) if dist is not None else None
    if args.l1_gather and args.dry_ranks:
        raise SystemExit("--l1-gather is a device-to-device RCCL all-gather: not part of a --dry-ranks run")
    if args.l1_gather and books_total != world * B:
        raise SystemExit("--l1-gather needs equal shards (all_gather_into_tensor)")
    l1 = parallel.L1Gather(env, dist) if (dist is not None and args.l1_gather) else None

    def run_steps(n):
        done = 0
        while done < n:
            c = min(spl, n - done)
            env.clear_history()   # the consumer has drained the previous launch's records
            env.clear_trades()
            env.run(c, sync=False)
            if gather is not None:
                gather.all_gather()   # 64 B per GPU over RCCL; never on the stepping critical path
            if l1 is not None:
                l1.all_gather()       # optional tier: 36 B per book, queued behind the launch on the same stream
            done += c

    # Pre-heat: the GPU's clocks fall within milliseconds of idling and need sustained load to come back (k_agents_fsm, a pure
    # latency chain, runs 170 us per launch cold and 155 us warm: scripts/region_trace.sh), and the driver's command line
    # (--warmup 5 = 1.7 ms of work after seconds of host-side set-up) would time the ramp.  bk_warm (a public entry of the
    # library, include/bourse_amd.h) steps THIS env's books with its own kernels and puts the state back.  How much load
    # the ramp needs differs from box to box (round 4: 100 steps = 22 ms were enough on the builder's boxes, not on the
    # driver's, whose regions climbed 272 -> 294 M), so the pre-heat is ADAPTIVE: chunks of --preheat-steps scratch steps
    # until the last three chunk rates agree within 1 % AND --preheat-min-ms have passed (at most --preheat-max-chunks), and the
    # line reports what ran.
    preheat_chunk, preheat_ran = [max(1, args.preheat_steps)], [0]

    def preheat(first):
        if args.preheat_steps <= 0:
            return []
        rates = []
        began = time.perf_counter()
        while len(rates) < max(1, args.preheat_max_chunks):
            torch.cuda.synchronize()
            t = time.perf_counter()
            env.warm(preheat_chunk[0])
            torch.cuda.synchronize()
            d = time.perf_counter() - t
            preheat_ran[0] += preheat_chunk[0]
            rates.append(B * preheat_chunk[0] / d)
            if d < 8e-3 and args.preheat_max_chunks > 1:  # a chunk is >= ~10 ms of load whatever the batch size
                preheat_chunk[0] = int(min(4000, max(preheat_chunk[0] + 1, preheat_chunk[0] * 10e-3 / d)))
            # (every rank stops on its own clock: the ranks share nothing, and the barrier in front of t0 lines them up)
            long_enough = not first or args.preheat_max_chunks <= 1 or (time.perf_counter() - began) * 1e3 >= args.preheat_min_ms
            if len(rates) >= (3 if first else 2) and max(rates[-3:]) / min(rates[-3:]) < 1.01 and long_enough:
                break
        return rates

    # Everything that reads the device or allocates happens BEFORE the pre-heat, so that nothing idles the GPU between it
    # and t0: the counters (bk_warm leaves them alone: its scratch steps write no trade record and the order counter is
    # part of the state it restores), their staging buffers, the event pool and its calibration, RCCL's communicator
    # (built on the first collective, ~20 ms).
    tc0 = int(env.trade_counts().sum())
    oc0 = int(env.order_counts().sum())
    env.profile(args.profile_every), env.profile(False)
    if dist is not None:
        if gather is not None:
            gather.all_gather()
        dist.barrier()
        torch.cuda.synchronize()
    preheat_rates = preheat(True)
    if dist is not None and preheat_rates:  # ranks converge after different numbers of chunks: line them up, then one more
        dist.barrier()                      # chunk on every rank, so that none enters the warm-up from an idle wait
        env.warm(preheat_chunk[0])
        preheat_ran[0] += preheat_chunk[0]
    preheat_first_steps = preheat_ran[0]
    run_steps(args.warmup)
    env.profile(args.profile_every)  # (host-side switch: its event pool exists since the pre-heat)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0  # this rank's region; the job's is the slowest rank's (max below): no collective inside it
    if dist is not None:
        dist.barrier()
    env.profile(False)
    kind0 = "k_run_mixed" if mixed else ("k_run_wave" if pipe == "wave" else "k_run_random")
    kind1 = ("k_agents_mixed_wave" if pipe == "wave_split" else "k_agents_mixed_lanes") if mixed else (
        "k_agents_wave" if pipe == "wave_split" else "k_agents_fsm")
    KINDS = (kind0, kind1, "k_step_batch", "k_step_events")  # the library's profile slots, in its order
    per_kind = {k: env.profile_read_kind(i) for i, k in enumerate(KINDS)}
    env.profile_read()
    dt_ranks = all_ranks(dt)  # every rank's wall time of the region: a straggler is visible in the line
    dt = max(dt_ranks)

    flags = env.flags()
    if flags.any():
        raise SystemExit(f"device flags set during the timed region: {np.unique(flags)} (trade/history capacity?)")
    n_trades = int(env.trade_counts().sum()) - tc0
    n_new = int(env.order_counts().sum()) - oc0  # orders created = New events of warm-up + timed region (device counters)
    st = env.stats()
    sampled_in = "the timed region"
    if not any(nl for _, nl in per_kind.values()):
        # --profile-every 0 (an unperturbed timed region, e.g. under rocprofv3 --pmc): the launch durations of the roofline
        # come from a few sampled steps AFTER it
        torch.cuda.synchronize()
        env.profile(1)
        run_steps(min(8, args.steps))
        torch.cuda.synchronize()
        env.profile(False)
        per_kind = {k: env.profile_read_kind(i) for i, k in enumerate(KINDS)}
        env.profile_read()
        sampled_in = "%d steps after the timed region (--profile-every 0)" % min(8, args.steps)

    value = books_total * args.steps / dt  # whole job: every rank's books (shards differ by at most one book)
    # Roofline accounting: algorithmic HBM bytes per book-step of every step kernel, in the
    # device layout actually shipped (S = per-book state block, 32 B trade records, measured event/trade rates):
    #   k_run_random / k_run_mixed (fused, spl steps per launch): 2 S / spl + L2 record + 32 N_tr
    #   k_agents_fsm  (one part, one step): 32 B in (RNG + live masks), 16 (RNG) + 4 (N_ev) + 2 N_ev + 8 N_new out
    #   k_step_batch  (one part, one step): 2 S + batch in (64 + 2 N_ev + 8 N_new) + L2 record + 32 N_tr
    S = env.state_bytes_per_book()
    W4 = env.width * 4
    tr_per_bs = n_trades / (B * (args.steps + args.warmup))
    ev_per_bs = st["sum_events"] / (B * (args.steps + args.warmup))
    new_per_bs = n_new / (B * (args.steps + args.warmup))
    acct = {"S": S, "W4": W4, "ev": ev_per_bs, "tr": tr_per_bs, "new": new_per_bs, "spl": spl, "pipe": pipe}
    per_bs, l2_bs = {}, {}
    for k in (kind0, "k_agents_fsm", "k_agents_wave", "k_agents_mixed_lanes", "k_agents_mixed_wave", "k_step_batch", "k_step_events"):
        per_bs[k], l2_bs[k] = algorithmic_bytes(k, **acct)
    bs_per_launch = {kind0: B * spl, "k_agents_fsm": B / parts, "k_agents_wave": B / parts, "k_agents_mixed_lanes": B / parts,
                     "k_agents_mixed_wave": B / parts,
                     "k_step_batch": B / parts,
                     "k_step_events": B}
    # PMC figures (HBM traffic, instruction counts) cannot be collected inside this process: they are rocprofv3 --pmc
    # passes of this same command (scripts/profile_round.sh), committed under profiles/ and REPLAYED here, keyed by
    # workload, and only when this run has the profiled run's shape (books per launch): `traffic_source` says so.
    pmc, pmc_src = {}, None
    try:
        allp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        # keyed by workload AND books per GPU: per-book-step PMC figures of one batch size do not carry over to another
        key = f"{args.workload}/{B}"
        pmc = allp.get(key, {})
        pmc_src = ("profiles/pmc_traffic.json[%s]: %s; replayed, not measured in this run" % (key, allp.get("_source", "")) if pmc else None)
    except Exception:
        pass
    kernels = {}
    for k, (ms, nl) in per_kind.items():
        if not nl:
            continue
        avg = ms / nl
        ach = per_bs[k] * bs_per_launch[k] / (avg * 1e-3) / 1e9
        tb = pmc.get(k if k != "k_run_mixed" else "k_run_random", {}).get("hbm_bytes_per_book_step")
        kernels[k] = {"avg_launch_ms": avg, "launches": nl, "bytes_per_book_step": per_bs[k],
                      "l2_bytes_per_book_step": l2_bs[k],
                      "book_steps_per_launch": bs_per_launch[k], "achieved": ach, "frac": ach / HBM_PEAK_GBPS,
                      "traffic": tb * bs_per_launch[k] if tb is not None else None}
    # the roofline kernel is the one that moves the bytes (the HBM roofline is about bytes); in the split pipeline the
    # lane-per-book k_agents_fsm is latency-bound, overlapped, and touches ~0.2 KB per book-step
    dominant = max(kernels, key=lambda k: kernels[k]["bytes_per_book_step"] * kernels[k]["book_steps_per_launch"] * kernels[k]["launches"])
    K = kernels[dominant]
    achieved, avg_ms, n_launch, traffic = K["achieved"], K["avg_launch_ms"], K["launches"], K["traffic"]
    bytes_per_bookstep, book_steps_per_launch = K["bytes_per_book_step"], K["book_steps_per_launch"]
    # What actually bounds the path: instruction ISSUE, not bytes.  The ports' peaks are MEASURED ones
    # (scripts/micro/mixed_issue_bench.hip, 8 waves per SIMD, independent instructions): s_add_u32 alone 0.95 per CU and
    # clock, v_add_u32 alone 1.65, a 1:1 mix of both 1.70 in total (0.85 + 0.85: the scalar port binds a balanced mix).
    # The counts per book-step are PMC figures committed under profiles/ (scalar = SALU + branches: one port).
    issue = None
    SCALAR_PEAK, VECTOR_PEAK, MIX_1TO1_PEAK = 0.95, 1.65, 1.70
    ins = {k: pmc.get(k, {}).get("insts_per_book_step") for k in kernels}
    if all(ins.values()) and ins:
        salu = sum(v["salu"] for v in ins.values())
        valu = sum(v["valu"] for v in ins.values())
        branch = sum(v.get("branch", 0.0) for v in ins.values())
        cu_clk = N_CU * CLOCK_GHZ * 1e9  # CU-clocks per second, whole GPU
        per_gpu = value / world
        s_rate, v_rate = (salu + branch) * per_gpu / cu_clk, valu * per_gpu / cu_clk  # instructions per CU and clock
        issue = {"bound": "instruction issue: one scalar port (SALU + branches) and one vector port per CU",
                 "salu_per_book_step": salu, "branch_per_book_step": branch, "valu_per_book_step": valu,
                 "scalar_per_cu_clk": s_rate, "vector_per_cu_clk": v_rate,
                 "scalar_peak_per_clk": SCALAR_PEAK, "vector_peak_per_clk": VECTOR_PEAK, "mixed_1to1_total_peak_per_clk": MIX_1TO1_PEAK,
                 "peaks_source": "scripts/micro/mixed_issue_bench.hip on this GPU model (profiles/r04/mixed_issue_bench.txt): "
                                 "s_add_u32 0.95, v_add_u32 1.65, 1:1 mix 1.70 instructions per CU and clock at 8 waves per SIMD",
                 "scalar_port_frac": s_rate / SCALAR_PEAK, "vector_port_frac": v_rate / VECTOR_PEAK,
                 "total_per_cu_clk": s_rate + v_rate,
                 "events_per_s_per_cu": ev_per_bs * per_gpu / N_CU, "assumed_clock_ghz": CLOCK_GHZ, "n_cu": N_CU}
        occ = pmc.get(dominant, {}).get("occupancy")
        if occ:  # wave-cycle split of the dominant kernel (PMC, committed under profiles/)
            issue["occupancy"] = occ
    out = {
        "metric": "book-steps/sec", "value": value, "unit": "book-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u32", "data": "synthetic", "preheat_steps": preheat_ran[0],
        "rank_region_ms": {"min": min(dt_ranks) * 1e3, "median": float(np.median(dt_ranks)) * 1e3, "max": max(dt_ranks) * 1e3,
                           "all": [d * 1e3 for d in dt_ranks]},
        "config": {
            "workload": f"{args.workload}: {books_total} books ({B}/GPU) x {n_agents} on-device agents "
                        f"({len(groups)} {'members: ' + '+'.join(g[0] for g in groups) if mixed else 'groups'}), {levels} levels/side, tick {TICK}, step_size {STEP_SIZE}, "
                        f"seed {SEED}+book",
            "books_total": books_total, "books_per_gpu": B, "ranks": world, "agents_per_book": n_agents, "levels": levels,
            "steps_per_launch": spl,
            "preheat": (f"bk_warm: scratch steps of this env right before the warm-up and before each repeated region, state "
                        f"restored (clock ramp), in chunks until three (repeats: two) chunk rates agree within 1 %; before the "
                        f"reported region: {len(preheat_rates)} chunks, {preheat_first_steps} steps; "
                        f"simulated: {args.warmup} warm-up + {args.steps} timed steps") if args.preheat_steps > 0 else "none",
            "preheat_chunk_rates": preheat_rates,
            "parallelism": f"{books_total} books in {world} contiguous shards ({args.scaling} scaling), no data-path "
                           f"collective, 64 B stats all-gather per launch" if world > 1 else "single GPU",
            "trades_per_book_step": tr_per_bs, "events_per_book_step": ev_per_bs,
            "events_per_s": ev_per_bs * value, "trades_per_s": tr_per_bs * value,
            "pipeline": f"{pipe} ({kind1} + k_step_batch per step, {parts} book parts on separate streams)"
            if pipe in ("split", "wave_split") else f"fused ({kind0})",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": dominant,
            "avg_launch_ms": avg_ms, "launches": n_launch, "bytes_per_book_step": bytes_per_bookstep,
            "book_steps_per_launch": book_steps_per_launch,
            "kernels": kernels,
            "accounting": acct,
            "issue": issue,
            "launches_sampled_in": sampled_in,
        },
    }
    # median of >= 5 runs.  `value` above is the contract's single timed region; the same region is repeated
    # (same barriers, max over ranks) and all values are listed beside it.
    vals = [value]
    for _ in range(max(0, args.repeats)):
        if preheat(False) and dist is not None:  # (the accounting above read the device: same footing as the reported region)
            dist.barrier()
            env.warm(preheat_chunk[0])
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        d = max(all_ranks(d))
        vals.append(books_total * args.steps / d)
    out["runs"] = {"n": len(vals), "values": vals, "median": float(np.median(vals))}
    # The per-launch roofline above is measured while the parts' kernels overlap each other on separate streams, which
    # stretches every launch.  Two unambiguous figures beside it: (1) aggregate = the algorithmic bytes of ALL step
    # kernels per step / the step's wall time; (2) the dominant kernel launched ALONE over the whole batch (one part).
    R = out["roofline"]
    step_bytes = sum(kernels[k]["bytes_per_book_step"] for k in kernels) * B  # every kernel visits every book once per step
    R["achieved_node"] = step_bytes * world / (out["ms_per_step"] * 1e-3) / 1e9
    R["peak_node"] = HBM_PEAK_GBPS * world
    R["frac_node"] = R["achieved_node"] / R["peak_node"]
    R["traffic_source"] = pmc_src if traffic is not None else None
    if pipe in ("split", "wave_split") and parts > 1:
        parts_set = env.split_parts()  # (restored below: BOURSE_AMD_SPLIT_PARTS / BOURSE_AMD_MIN_PART may have set them)
        env.set_split_parts(1, 64)
        if pipe == "wave_split":
            env.set_wave_options(64, 1)
        env.profile(1)
        run_steps(min(16, args.steps))
        env.sync()
        env.profile(False)
        ms1, n1 = env.profile_read_kind(2)
        env.profile_read()
        env.set_split_parts(*parts_set)
        if pipe == "wave_split":
            env.set_wave_options(64, args.wave_parts)
        if n1:
            a1 = per_bs["k_step_batch"] * B / (ms1 / n1 * 1e-3) / 1e9
            R["standalone"] = {"kernel": "k_step_batch", "book_steps_per_launch": B, "avg_launch_ms": ms1 / n1, "launches": n1,
                               "achieved": a1, "frac": a1 / HBM_PEAK_GBPS,
                               "note": "one launch over the whole batch, nothing overlapping it (k_agents_fsm runs before it)"}
    flags = env.flags()
    if flags.any():
        raise SystemExit(f"device flags set during the repeated regions: {np.unique(flags)}")
    if dist is not None:
        # every rank's shard, pipeline and part count (strong scaling cuts the workload into shards that may take different
        # pipelines than the one-GPU run: 8 192 books run wave_split where 65 536 run split)
        pipes = ("fused", "split", "wave_split", "wave")
        rp, rn, rb, rf = (all_ranks(float(pipes.index(pipe))), all_ranks(float(parts)), all_ranks(float(B)), all_ranks(float(first_book)))
        out["config"]["ranks_detail"] = [{"rank": r, "first_book": int(rf[r]), "books": int(rb[r]), "pipeline": pipes[int(rp[r])],
                                          "parts": int(rn[r])} for r in range(world)]
        out["config"]["dist_backend"] = dist_backend
    if gather is not None:
        g = gather.result()
        out["config"]["stats_allgather"] = {"zero_copy": gather.zero_copy, "ranks_seen": int(gather.out.shape[0]),
                                            "n_books": g["n_books"], "sum_trades": g["sum_trades"]}
        if g["n_books"] != books_total or gather.out.shape[0] != world:
            raise SystemExit(f"stats all-gather inconsistent: {g}")
    if l1 is not None:
        rec = l1.result()
        mine = env.level2()[:, :9]
        if rec.shape != (world * B, 9) or not np.array_equal(rec[rank * B:(rank + 1) * B], mine):  # equal shards only
            raise SystemExit("L1 all-gather inconsistent with this rank's level-2 records")
        out["config"]["l1_allgather"] = {"bytes_per_gpu": int(B * 36), "books": int(rec.shape[0])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(groups, levels, B)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        rank_watchdog(0)
    if args.dry_ranks:
        out["dry_ranks"] = True
        out["config"]["parallelism"] += " - DRY RUN: all ranks on ONE GPU, collectives over gloo; the rate is not an N-GPU measurement"
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
