#!/usr/bin/env python3
"""How fast can the host enqueue step launches (empty queue, no waiting)?  If this is close to the device time per
launch the benchmark is host-bound.  GPU box."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
eng.reset(obs=out[0])
for _ in range(300):
    eng.step_random(out=out)
torch.cuda.synchronize()
for n in (50, 200, 1000):
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.step_random(out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) * 1e6 / n, (t2 - t0) * 1e6 / n))
    print("n=%d: host enqueue us/launch %s | enqueue+drain us/launch %s" % (n, ["%.2f" % a for a, _ in res], ["%.2f" % b for _, b in res]))
# the raw C call without the Python wrapper layers
L, h, st, dp = eng._L, eng._h, eng._stream(), eng._dp
o, r, d = dp(out[0]), dp(out[1]), dp(out[2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    L.ssd_step_random(h, eng.num_actions, None, o, r, d, 0, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("raw ctypes call: %.2f us/launch" % ((t1 - t0) * 1e6 / 200))

# one library call for the whole rollout
ring = 4
ro = torch.empty((ring,) + tuple(out[0].shape), dtype=torch.uint8, device="cuda")
rr = torch.empty((ring,) + tuple(out[1].shape), dtype=torch.int32, device="cuda")
rd = torch.empty((ring,) + tuple(out[2].shape), dtype=torch.uint8, device="cuda")
for n in (1000, 3000):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(n, ro, rr, rd, reset_every=1000)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("rollout_random(%d): host %.2f us/launch, total %.2f us/launch" % (n, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6 / n))

# the fused rollout kernel: ONE launch for all n steps
for n in (1000, 3000):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(n, ro, rr, rd, reset_every=1000, fused=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("rollout_random(%d, fused): %.2f us/step = %.2f G agent-steps/s" % (n, (t2 - t0) * 1e6 / n, 4096 * 5 * n / (t2 - t0) / 1e9))
