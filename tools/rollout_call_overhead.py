import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs(); ring = tuple(t.unsqueeze(0) for t in out)
eng.rollout_random(500, *ring, reset_every=1000); torch.cuda.synchronize()
for n in (100, 1000, 3000, 10000):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.rollout_random(n, *ring, reset_every=1000)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("n=%5d: call returned after %.2f ms, done after %.2f ms = %.2f us/step" % (n, (t1-t0)*1e3, (t2-t0)*1e3, (t2-t0)*1e6/n))
