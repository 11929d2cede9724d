#!/usr/bin/env python3
"""What an RCCL barrier just ahead of a short rollout call costs that call (1 rank under torch.distributed.run):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29520 tools/rccl_short_call_probe.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from sequential_social_dilemma_games_amd import constants as K, parallel
from sequential_social_dilemma_games_amd.engine import VecEngine

dist, rank, world, local_rank = parallel.init_process_group()
torch.cuda.set_device(local_rank)
eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0, device=local_rank)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.set_rollout_chains(2)
eng.rollout_random(20, *ring, reset_every=1000, step0=0); torch.cuda.synchronize()

def timed(label, before):
    xs = []
    for rep in range(8):
        before()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.rollout_random(20, *ring, reset_every=1000, step0=1); torch.cuda.synchronize()
        xs.append((time.perf_counter() - t0) * 1e6)
    print("%-46s %s" % (label, " ".join("%.0f" % x for x in xs)), flush=True)

timed("no collective yet", lambda: None)
timed("barrier before every call (first = comm init)", lambda: parallel.barrier(dist, local_rank))
timed("no barrier (after RCCL is up)", lambda: None)
timed("barrier, then 200 us of sleep", lambda: (parallel.barrier(dist, local_rank), torch.cuda.synchronize(), time.sleep(0.0002)))
timed("barrier, then a 4-step call", lambda: (parallel.barrier(dist, local_rank), torch.cuda.synchronize(), eng.rollout_random(4, *ring, reset_every=1000, step0=1)))
t = torch.zeros(1, device="cuda")
timed("all_reduce on the current stream before", lambda: dist.all_reduce(t))
