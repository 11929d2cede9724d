import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

def t(eng, ring, label):
    for fused in (False, True):
        eng.rollout_random(300, *ring, reset_every=1000, fused=fused)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.rollout_random(2000, *ring, reset_every=1000, fused=fused)
        b.record()
        torch.cuda.synchronize()
        print("%-50s %s %.2f us/step" % (label, "fused   " if fused else "per-step", a.elapsed_time(b) * 1e3 / 2000), flush=True)

mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
if mode in ("nccl_first", "nccl_first_barrier", "nccl_first_devid"):
    if mode == "nccl_first_devid":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    if mode == "nccl_first_barrier":
        dist.barrier()
eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(x.unsqueeze(0) for x in out)
t(eng, ring, mode + ": engine created now")
if mode == "engine_first":
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.barrier()
    t(eng, ring, mode + ": after nccl init + barrier")
    eng2 = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
    out2 = eng2.alloc_outputs(); ring2 = tuple(x.unsqueeze(0) for x in out2)
    t(eng2, ring2, mode + ": SECOND engine created after nccl")
else:
    dist.barrier()
    t(eng, ring, mode + ": after a barrier")
