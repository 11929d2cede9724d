#!/usr/bin/env python3
"""The fused rollout kernel in short and long calls (4096 Harvest envs): us per step of n-step calls, median of 15.
    [SSD_LIB_PATH=...] python tools/fused_short.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.reset(obs=out[0]); torch.cuda.synchronize()
print("lib:", os.environ.get("SSD_LIB_PATH", "(product)"))
for n in (5, 20, 50, 200, 1000):
    xs = []
    for rep in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.rollout_random(n, *ring, reset_every=1000, step0=1, fused=True); torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6)
    xs.sort()
    print("fused n=%4d: %.1f us = %.2f us/step" % (n, xs[len(xs) // 2], xs[len(xs) // 2] / n))
