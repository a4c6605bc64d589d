"""What does k_agents_fsm lose to the re-convergence of its 64 lanes at a segment boundary?  
65 536 books x ONE RandomAgents group of 128 agents (so that a free-running lane needs no per-lane group parameters - the best
case), split pipeline; the shipped kernel re-converges after agent 63, a -DBOURSE_AMD_FSM_FREERUN=1 build does not.
usage (GPU box): BOURSE_AMD_LIBRARY=<variant .so> python scripts/fsm_freerun_probe.py      prints rate + kernel times + a state hash"""
import hashlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bourse_amd as bk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
groups = [(128, (32, 64), (10, 20), 2, 0.5)]
env = bk.ManyBookEnv(B, 101, 0, 2, 100_000, True, levels=32, max_live_orders=128, trade_capacity=96 * 50, history_capacity=50,
                     stream=torch.cuda.current_stream().cuda_stream, strict=False)
env.set_random_agents(groups)
env.set_pipeline("split")
for _ in range(3):
    env.clear_history(); env.clear_trades(); env.run(50)
torch.cuda.synchronize()
vals = []
for rep in range(5):
    env.clear_history(); env.clear_trades()
    env.profile(8 if rep == 4 else 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(4):
        env.clear_history(); env.clear_trades(); env.run(50)
    torch.cuda.synchronize(); vals.append(B * 200 / (time.perf_counter() - t))
fsm, step = env.profile_read_kind(1), env.profile_read_kind(2)
env.profile_read(); env.profile(0)
h = hashlib.sha1(env.level2().tobytes()).hexdigest()[:12]
print("lib %s  books %d  %s M book-steps/s (median %.1f)  k_agents_fsm %.1f us  k_step_batch %.1f us  flags %s  l2 hash %s" % (
    os.path.basename(os.environ.get("BOURSE_AMD_LIBRARY", "in-tree")), B, [round(v / 1e6, 1) for v in vals], float(np.median(vals)) / 1e6,
    1e3 * fsm[0] / max(fsm[1], 1), 1e3 * step[0] / max(step[1], 1), np.unique(env.flags()), h))
env.close()
