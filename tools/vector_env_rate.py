#!/usr/bin/env python3
"""SSDVectorEnv.step throughput (policy-driven stepping through the batched adapter, 4096 envs): with the envs in sync
(reset launch on the horizon step only) and out of sync (a masked reset launch after every step).  GPU box."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.vector_env import SSDVectorEnv
for sync in (True, False):
    v = SSDVectorEnv(K.GAME_HARVEST, 4096, 5, horizon=1000)
    v.reset()
    if not sync:
        v.engine.steps_since_full_reset = None
    acts = torch.randint(0, 8, (4096, 5), dtype=torch.int32, device="cuda")
    for _ in range(300): v.step(acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3000): v.step(acts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("SSDVectorEnv.step (policy actions, auto-reset %s): %.2f us/step = %.2f G agent-steps/s" % ("on the horizon step only" if sync else "launch every step", dt * 1e6 / 3000, 4096 * 5 * 3000 / dt / 1e9))
