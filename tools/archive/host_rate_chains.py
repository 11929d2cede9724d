#!/usr/bin/env python3
"""Host launch capacity of ssd_rollout_random by number of chains: with 64 envs the kernels are tiny, so the time per
step is what the host threads can enqueue (GPU box)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

for E in (64, 4096):
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    ring = tuple(t.unsqueeze(0) for t in out)
    for chains in (1, 2, 3, 4):
        eng.set_rollout_chains(chains)
        eng.rollout_random(300, *ring)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(3000, *ring)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("E=%d chains=%d: host %.2f us/step, total %.2f us/step" % (E, chains, (t1 - t0) * 1e6 / 3000, (t2 - t0) * 1e6 / 3000))
