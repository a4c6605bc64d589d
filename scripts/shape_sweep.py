"""bk_run pipelines vs. batch size for every BASELINE agent shape (GPU box): the data behind the auto rule.
usage: python scripts/shape_sweep.py [shapes] [sizes] [pipelines] > table; writes gpurun_out/shape_sweep.json
shapes: C2 (64 RandomAgents, 64-slot pool, 16 levels), C3 (128, 128-slot, 32), C5 (512, 512-slot, 64),
C5M (256 momentum + 256 noise, 512-slot, 64)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bourse_amd

MOM = dict(tick_size=2, p_cancel=0.1, trade_vol=100, decay=1.0, demand=20.0, scale=0.5, order_ratio=1.0, price_dist_mu=0.0, price_dist_sigma=10.0)
NOISE = dict(tick_size=2, p_limit=0.3, p_market=0.2, p_cancel=0.2, trade_vol=100, price_dist_mu=0.0, price_dist_sigma=1.0)
SHAPES = {
    "C2": (16, 64, [(32, (40, 56), (10, 20), 2, 0.8), (32, (40, 56), (50, 70), 2, 0.2)], None),
    "C3": (32, 128, [(64, (32, 64), (10, 20), 2, 0.8), (64, (32, 64), (50, 70), 2, 0.2)], None),
    "R4": (32, 256, [(128, (32, 64), (10, 20), 2, 0.8), (128, (32, 64), (50, 70), 2, 0.2)], None),  # a 256-slot pool
    "C5": (64, 512, [(256, (100, 164), (10, 20), 2, 0.8), (256, (100, 164), (50, 70), 2, 0.2)], None),
    "C5M": (64, 512, None, [("momentum", 0, 256, MOM), ("noise", 256, 256, NOISE)]),
}
shapes = (sys.argv[1] if len(sys.argv) > 1 else "C2,C3,C5,C5M").split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,2048,4096,6144,8192,12288,16384,24576,32768,65536").split(",")]
pipes_arg = sys.argv[3].split(",") if len(sys.argv) > 3 else None
T = 30
out = {}
for name in shapes:
    levels, pool, groups, members = SHAPES[name]
    pipes = pipes_arg or (["fused", "split", "wave_split", "wave", "auto"] if groups else ["fused", "split", "wave_split", "auto"])
    n_agents = pool
    for B in sizes:
        if pool == 512 and B > 32768:
            continue
        row = {}
        for pipe in pipes:
            if pipe == "fused" and B > 16384:
                continue  # (far behind there, and slow to run)
            env = bourse_amd.ManyBookEnv(B, 101, 0, 2, 100_000, levels=levels, max_live_orders=pool,
                                         trade_capacity=max(64, n_agents // 2) * T, history_capacity=T, strict=False)
            if groups:
                env.set_random_agents(groups)
            else:
                env.set_agents(members)
            env.set_pipeline(pipe)
            env.run(T); env.clear_trades()
            best = 0.0
            for rep in range(3):
                env.clear_history(); env.clear_trades()
                t0 = time.perf_counter(); env.run(T); dt = time.perf_counter() - t0
                best = max(best, B * T / dt / 1e6)
            row[pipe] = round(best, 2)
            if pipe == "auto":
                row["auto_is"] = "%s/%d" % env.pipeline()
            env.close()
        out.setdefault(name, {})[str(B)] = row
        winner = max((k for k in row if k not in ("auto", "auto_is")), key=lambda k: row[k])
        print(f"{name:4s} B={B:6d}  " + "  ".join(f"{k} {v}" for k, v in row.items()) + f"   best={winner} auto/best={row.get('auto', 0) / row[winner]:.3f}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", "shape_sweep.json")
try:
    old = json.load(open(path))
except Exception:
    old = {}
old.update(out)
json.dump(old, open(path, "w"), indent=1)
