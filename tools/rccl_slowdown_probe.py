#!/usr/bin/env python3
"""Does creating an RCCL communicator change the step kernel's speed?  (GPU box)  Times the fused rollout before / after
init_process_group('nccl'), after the first collective, and after destroy_process_group()."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)


def t(label):
    for fused in (False, True):
        eng.rollout_random(300, *ring, reset_every=1000, fused=fused)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.rollout_random(2000, *ring, reset_every=1000, fused=fused)
        b.record()
        torch.cuda.synchronize()
        print("%-40s %s %.2f us/step" % (label, "fused   " if fused else "per-step", a.elapsed_time(b) * 1e3 / 2000), flush=True)


t("before any process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
t("gloo group")
dist.barrier()
t("gloo group + barrier")
dist.destroy_process_group()
os.environ["MASTER_PORT"] = "29534"
dist.init_process_group("nccl", rank=0, world_size=1)
t("nccl group, no collective yet")
x = torch.zeros(1, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
t("nccl group after an all_reduce")
dist.destroy_process_group()
t("after destroy_process_group")
