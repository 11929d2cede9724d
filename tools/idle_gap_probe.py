#!/usr/bin/env python3
"""What does an idle gap ahead of a 20-step rollout call cost?  (As a rank under torch.distributed.run the driver's timed call follows
an RCCL barrier -- about a millisecond in which nothing of this process runs on the GPU -- and takes 143 us instead of 131.)
us per 20-step call after a pause of 0 / 0.1 / 1 / 5 ms; GPU_MAX_HW_QUEUES as set by the caller.   python tools/idle_gap_probe.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
from _label import label
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

label("idle_gap_probe")
eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.rollout_random(200, *ring, reset_every=1000, step0=0); torch.cuda.synchronize()
x = torch.zeros(1 << 20, device="cuda")
for gap_ms, busy in ((0, False), (0.1, False), (1, False), (5, False), (1, True)):
    xs = []
    for rep in range(15):
        eng.rollout_random(5, *ring, reset_every=1000, step0=200); torch.cuda.synchronize()
        if busy:                                   # the gap filled with small kernels of the process's own (the GPU never idles)
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap_ms * 1e-3:
                x.add_(1.0)
            torch.cuda.synchronize()
        elif gap_ms:
            time.sleep(gap_ms * 1e-3)
        t0 = time.perf_counter(); eng.rollout_random(20, *ring, reset_every=1000, step0=205); torch.cuda.synchronize()
        xs.append((time.perf_counter() - t0) * 1e6)
    xs.sort()
    print("pause %4.1f ms%s: 20-step call %.1f us (min %.1f max %.1f) = %.2f us per step" % (gap_ms, " (filled with small kernels)" if busy else "", xs[7], xs[0], xs[-1], xs[7] / 20))
