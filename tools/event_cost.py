"""tools/event_cost.py -- GPU box: what a torch.cuda.Event.record() between a 20-step rollout call and the closing synchronize adds
to the call's wall time (bench.py records its closing HIP event there).  Alternating, medians of 40 calls each."""
import sys, time, statistics
sys.path.insert(0, ".")
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

eng = VecEngine(K.GAME_HARVEST, K.HARVEST_MAP, num_envs=4096, num_agents=5, seed=0)
eng.reset()
ring = (torch.zeros((1, 4096, 5, 15, 15, 3), dtype=torch.uint8, device="cuda"), torch.zeros((1, 4096, 5), dtype=torch.int32, device="cuda"),
        torch.zeros((1, 4096, 5), dtype=torch.uint8, device="cuda"))
eng.set_rollout_chains(2)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {"plain": [], "event_after_call": [], "event_before_and_after": []}
s0 = 0
for i in range(8):
    eng.rollout_random(20, *ring, step0=s0); s0 += 20
torch.cuda.synchronize()
for rep in range(40):
    for kind in res:
        if kind == "event_before_and_after":
            ev0.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(20, *ring, step0=s0)
        if kind != "plain":
            ev1.record()
        torch.cuda.synchronize()
        res[kind].append((time.perf_counter() - t0) * 1e6)
        s0 += 20
for kind, v in res.items():
    print("%-24s median %.1f us  min %.1f  (20-step call + synchronize)" % (kind, statistics.median(v), min(v)))
eng.close()
