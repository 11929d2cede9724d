#!/usr/bin/env python3
"""One fused rollout (SSD_ROLLOUT_FUSED) of n steps -- the workload for rocprofv3 --pmc passes over the rollout kernel.
    python3 tools/fused_probe.py [harvest|cleanup] [E] [n_steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

game = K.GAME_CLEANUP if (len(sys.argv) > 1 and sys.argv[1] == "cleanup") else K.GAME_HARVEST
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
eng = VecEngine(game, None, num_envs=E, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.reset(obs=out[0])
for _ in range(3):
    eng.rollout_random(n, *ring, reset_every=1000, fused=True)
torch.cuda.synchronize()
