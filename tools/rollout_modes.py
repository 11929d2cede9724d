#!/usr/bin/env python3
"""Who should enqueue a rollout's chains: the calling thread (launches of all chains interleaved step by step) or the
persistent worker threads (one per extra chain)?  Wall time per step of short and long ssd_rollout_random calls, 4096 envs,
2 chains.  Run with SSD_ROLLOUT_THREADS=0 and =1 (the knob is read once per process)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.set_rollout_chains(chains)
eng.rollout_random(300, *ring, reset_every=1000)
torch.cuda.synchronize()
print("SSD_ROLLOUT_THREADS=%s E=%d chains=%d" % (os.environ.get("SSD_ROLLOUT_THREADS"), E, chains))
for n in (5, 20, 20, 20, 32, 100, 100, 1000, 3000):
    best = None
    for rep in range(3):
        eng.rollout_random(5, *ring, reset_every=1000, step0=1)
        torch.cuda.synchronize()
        time.sleep(0.002)
        t0 = time.perf_counter()
        eng.rollout_random(n, *ring, reset_every=1000, step0=1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = ((t2 - t0) * 1e6 / n, (t1 - t0) * 1e6 / n)
        best = r if best is None or r[0] < best[0] else best
    print("n=%5d: %.2f us/step wall (host enqueue %.2f us/step), total %.1f us" % (n, best[0], best[1], best[0] * n))
