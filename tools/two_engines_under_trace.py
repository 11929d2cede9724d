import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
which = sys.argv[1:] or ["harvest", "cleanup"]
for name in which:
    game, amap = (K.GAME_HARVEST, None) if name == "harvest" else (K.GAME_CLEANUP, None)
    eng = VecEngine(game, amap, num_envs=4096, num_agents=5, seed=0)
    out = eng.alloc_outputs(); ring = tuple(t.unsqueeze(0) for t in out)
    eng.rollout_random(20, *ring, reset_every=1000); torch.cuda.synchronize()
    print(name, eng.rollout_path(), flush=True)
    eng.close()
print("done")
