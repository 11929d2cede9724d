#!/usr/bin/env python3
"""Probe (GPU box): 4096 envs as one pipelined handle of EA envs plus one plain handle of 4096 - EA envs, stepped
concurrently from two host threads -- would a mixed split beat two plain chains at the headline size?"""
import os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

STEPS = 3000


def run(EA, pipe_a, pipe_b):
    EB = 4096 - EA
    engs = [VecEngine(K.GAME_HARVEST, None, num_envs=n, num_agents=5, seed=0, env_index_base=b) for n, b in ((EA, 0), (EB, EA)) if n > 0]
    streams = [torch.cuda.Stream() for _ in engs]
    bufs = []
    for e in engs:
        e.set_rollout_chains(1)
        bufs.append((torch.empty((2, e.E, 5, 15, 15, 3), dtype=torch.uint8, device="cuda"), torch.empty((2, e.E, 5), dtype=torch.int32, device="cuda"),
                     torch.empty((2, e.E, 5), dtype=torch.uint8, device="cuda")))

    def work(i, n, step0):
        with torch.cuda.stream(streams[i]):
            engs[i].rollout_random(n, *bufs[i], reset_every=1000, step0=step0, pipelined=(pipe_a, pipe_b)[i])
    for n, s0 in ((300, 0), (STEPS, 300)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, n, s0)) for i in range(len(engs))]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / STEPS * 1e6


for EA, pa, pb in ((2048, False, False), (2048, True, False), (2560, True, False), (2816, True, False), (3072, True, False), (2048, True, True)):
    print("EA=%d (pipelined=%s) + EB=%d (pipelined=%s): %.2f us per 4096-env step" % (EA, pa, 4096 - EA, pb, run(EA, pa, pb)), flush=True)
