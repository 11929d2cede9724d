#!/usr/bin/env python3
"""Parity of the kernels compiled for the enlarged maps (Harvest 25x38, Cleanup 48x36; FAST = 2): 2100 envs x 400 steps call by
call, as two chains and as the fused rollout kernel, against the oracle.  GPU box."""
import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import golden_util as G
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
for game, amap, N in ((K.GAME_HARVEST, K.harvest_map_25x38(), 5), (K.GAME_CLEANUP, K.cleanup_map_48x36(), 10)):
    for how in ("step", "chains", "fused"):
        E = 2100
        eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=77)
        ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=77)
        out = eng.alloc_outputs(); ring = tuple(t.unsqueeze(0) for t in out)
        ora.reset()
        steps = 400
        if how == "step":
            eng.reset(obs=out[0])
            for s in range(steps):
                if s and s % 150 == 0:
                    eng.reset(obs=out[0])
                eng.step_random(out=out)
        else:
            eng.set_rollout_chains(2 if how == "chains" else 1)
            eng.rollout_random(steps, *ring, reset_every=150, fused=(how == "fused"))
        for s in range(steps):
            if s and s % 150 == 0:
                ora.reset()
            _, o_obs, o_rew, _ = ora.step_random(want_obs=(s == steps - 1))
        assert np.array_equal(out[0].cpu().numpy(), o_obs) and np.array_equal(out[1].cpu().numpy(), o_rew), (game, how)
        a, b = eng.get_state(), ora.get_state()
        for k in ("world", "pos", "orient", "episode", "t"):
            assert np.array_equal(a[k], b[k]), (game, how, k)
        assert eng.status() == 0
        print("game %d, %s: 2100 envs x %d steps bit-exact" % (game, how, steps), flush=True)
