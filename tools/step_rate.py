#!/usr/bin/env python3
"""Per-call stepping rate: VecEngine.step(actions) back to back, device action tensor (us per step; 4096 Harvest / Cleanup envs), and
bit-exactness of the last step against the oracle.   [SSD_LIB_PATH=...] python tools/step_rate.py [E]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch
import golden_util as G
from _label import label
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

label("step_rate " + " ".join(sys.argv[1:]))
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for game, amap in ((K.GAME_HARVEST, K.HARVEST_MAP), (K.GAME_CLEANUP, K.CLEANUP_MAP)):
    eng = VecEngine(game, amap, num_envs=E, num_agents=5, seed=3)
    ora = pyoracle.Oracle(game, amap, E, 5, G.default_lut(), seed=3)
    out = eng.alloc_outputs()
    eng.reset(obs=out[0]); ora.reset()
    na = eng.num_actions
    acts = torch.randint(0, na, (64, E, 5), dtype=torch.int32, device="cuda")
    a_host = acts.cpu().numpy()
    for k in range(64):                                  # parity of the path being timed
        o, r, _ = eng.step(acts[k], out=out)
        o_obs, o_rew, _ = ora.step(a_host[k])
    assert np.array_equal(o.cpu().numpy(), o_obs) and np.array_equal(r.cpu().numpy(), o_rew), "per-call step differs from the oracle"
    xs = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(300):
            eng.step(acts[k & 63], out=out)
        torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6 / 300)
    xs.sort()
    print("game %d, %d envs: per-call step %.2f us (min %.2f max %.2f), bit-exact after 64 steps" % (game, E, xs[2], xs[0], xs[-1]))
