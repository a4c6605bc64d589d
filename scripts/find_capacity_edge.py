"""Find AgentSet configurations at the pool-capacity edge: the lane-per-book members' update (pipeline 'split') flags
BK_FLAG_POOL_OVERFLOW while the fused kernel fits.  Used to pick the regression case of the guarded auto pipeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bourse_amd as bk
B, T = 4096, 30
for n, pl, pm_, pc in ((110, 0.5, 0.3, 0.5), (120, 0.4, 0.3, 0.6), (100, 0.6, 0.4, 0.4), (90, 0.8, 0.5, 0.5), (125, 0.3, 0.2, 0.7),
                       (115, 0.45, 0.35, 0.55), (105, 0.55, 0.45, 0.45)):
    P = dict(tick_size=1, p_limit=pl, p_market=pm_, p_cancel=pc, trade_vol=10, price_dist_mu=0.0, price_dist_sigma=1.0)
    res = {}
    for pipe in ("split", "fused", "auto"):
        e = bk.ManyBookEnv(B, 11, 0, 1, 1_000_000, True, levels=8, max_live_orders=128, trade_capacity=128 * T, history_capacity=T,
                           strict=False)
        e.set_agents([("noise", 0, n, P)])
        e.set_pipeline(pipe)
        e.run(T)
        f = e.flags()
        res[pipe] = (int((f & 1).astype(bool).sum()), e.pipeline_fallbacks(), e.history())
    same = np.array_equal(res["auto"][2], res["fused"][2])
    print(f"n={n} p_limit={pl} p_market={pm_} p_cancel={pc}: overflow books split={res['split'][0]} fused={res['fused'][0]} "
          f"auto={res['auto'][0]} fallbacks={res['auto'][1]} auto==fused: {same}", flush=True)
