#!/usr/bin/env python3
"""What the pieces of bench.py's timed region cost for the driver's command (K = 20 after W = 5): event records, the rollout
call, the synchronize -- a fresh process each time, like the driver's."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
use_events = "--no-events" not in sys.argv
eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.set_rollout_chains(2)
eng.rollout_random(5, *ring, reset_every=1000, step0=0)
torch.cuda.synchronize()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
if use_events: ev0.record()
t1 = time.perf_counter()
eng.rollout_random(20, *ring, reset_every=1000, step0=5)
t2 = time.perf_counter()
if use_events: ev1.record()
t3 = time.perf_counter()
torch.cuda.synchronize()
t4 = time.perf_counter()
print("events=%d: ev0.record %.1f us, rollout call %.1f us, ev1.record %.1f us, synchronize %.1f us; total %.1f us = %.2f us/step%s"
      % (use_events, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6, (t4 - t0) * 1e6 / 20,
         ("; events say %.1f us" % (ev0.elapsed_time(ev1) * 1e3)) if use_events else ""))
# again, warm
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.rollout_random(20, *ring, reset_every=1000, step0=25)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("  repeat: call %.1f us, total %.1f us" % ((t2 - t0) * 1e6, (t4 - t0) * 1e6))
