#!/usr/bin/env python3
"""Where the fixed cost of a short ssd_rollout_random call goes (4096 Harvest envs): empty synchronize, one launch + synchronize,
20-step calls with 1 and 2 chains, host-return vs completion.  Environment knobs to try: HSA_ENABLE_INTERRUPT=0, SSD_ROLLOUT_THREADS."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

def med(xs):
    xs = sorted(xs); return xs[len(xs) // 2]

eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.reset(obs=out[0]); torch.cuda.synchronize()
print("env: HSA_ENABLE_INTERRUPT=%s SSD_ROLLOUT_THREADS=%s" % (os.environ.get("HSA_ENABLE_INTERRUPT"), os.environ.get("SSD_ROLLOUT_THREADS")))
xs = []
for _ in range(50):
    t0 = time.perf_counter(); torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6)
print("empty synchronize: %.1f us" % med(xs))
xs = []
for _ in range(30):
    time.sleep(0.001)
    t0 = time.perf_counter(); eng.step_random(out=out); t1 = time.perf_counter(); torch.cuda.synchronize(); xs.append(((time.perf_counter() - t0) * 1e6, (t1 - t0) * 1e6))
print("one 4096-env step + synchronize: %.1f us (call returns after %.1f)" % (med([x[0] for x in xs]), med([x[1] for x in xs])))
for chains in (1, 2):
    eng.set_rollout_chains(chains)
    for n in (1, 5, 20, 100):
        xs = []
        for rep in range(12):
            time.sleep(0.001)
            t0 = time.perf_counter(); eng.rollout_random(n, *ring, reset_every=1000, step0=1); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            xs.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6))
        print("chains=%d n=%3d: first call %.1f us (host %.1f); median of 12: %.1f us total = %.2f us/step, host returns after %.1f, tail after return %.1f"
              % (chains, n, xs[0][0], xs[0][1], med([x[0] for x in xs]), med([x[0] for x in xs]) / n, med([x[1] for x in xs]), med([x[0] - x[1] for x in xs])))
# events: device-side duration of a 20-step call
eng.set_rollout_chains(2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
xs = []
for rep in range(10):
    time.sleep(0.001)
    e0.record(); eng.rollout_random(20, *ring, reset_every=1000, step0=1); e1.record(); torch.cuda.synchronize(); xs.append(e0.elapsed_time(e1) * 1e3)
print("HIP events around a 20-step 2-chain call: %.1f us" % med(xs))
