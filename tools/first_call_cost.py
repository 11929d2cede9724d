"""tools/first_call_cost.py -- GPU box: does the FIRST rollout call of a given length cost more than the ones after it?
The driver's command (bench.py --steps 20 --warmup 5) times the first 20-step call of the process: warm-up calls of 4 and 1 steps
come before it.  Wall time (synchronize on both sides) of consecutive calls of the given lengths, several fresh handles."""
import sys, time
sys.path.insert(0, ".")
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

seq = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,1,20,20,20,20").split(",")]
for rep in range(4):
    eng = VecEngine(K.GAME_HARVEST, K.HARVEST_MAP, num_envs=4096, num_agents=5, seed=rep)
    eng.reset()
    ring = (torch.zeros((1, 4096, 5, 15, 15, 3), dtype=torch.uint8, device="cuda"), torch.zeros((1, 4096, 5), dtype=torch.int32, device="cuda"),
            torch.zeros((1, 4096, 5), dtype=torch.uint8, device="cuda"))
    eng.set_rollout_chains(2)
    torch.cuda.synchronize()
    out, s0 = [], 0
    for n in seq:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(n, *ring, step0=s0)
        torch.cuda.synchronize()
        out.append("%d steps: %.1f us" % (n, (time.perf_counter() - t0) * 1e6))
        s0 += n
    print("handle %d: " % rep + " | ".join(out))
    eng.close()
