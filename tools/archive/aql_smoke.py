#!/usr/bin/env python3
"""First contact of the library's own AQL dispatch path (ssd_aql.hip) with the GPU: a handful of small rollouts against the
oracle, each step of the way printed before it runs (a hang then shows where)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("SSD_AQL_VERBOSE", "1")
import numpy as np
import torch
from oracle import pyoracle
from sequential_social_dilemma_games_amd import config, constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

def log(*a):
    print(*a, flush=True)

lut = config.make_lut()
for game, amap, E, chains, ring, steps, every in ((K.GAME_HARVEST, K.HARVEST_MAP, 64, 1, 1, 3, 0), (K.GAME_HARVEST, K.HARVEST_MAP, 300, 2, 3, 25, 7),
                                                  (K.GAME_CLEANUP, K.CLEANUP_MAP, 4096, 2, 1, 40, 16), (K.GAME_HARVEST, K.HARVEST_MAP, 4096, 3, 2, 2100, 1000)):
    log("case: game %d E=%d chains=%d ring=%d steps=%d reset_every=%d" % (game, E, chains, ring, steps, every))
    eng = VecEngine(game, amap, num_envs=E, num_agents=5, seed=3)
    ora = pyoracle.Oracle(game, amap, E, 5, lut, seed=3)
    eng.set_rollout_chains(chains)
    obs = torch.zeros((ring, E, 5, 15, 15, 3), dtype=torch.uint8, device="cuda")
    rew = torch.zeros((ring, E, 5), dtype=torch.int32, device="cuda")
    done = torch.zeros((ring, E, 5), dtype=torch.uint8, device="cuda")
    if every == 0:
        eng.reset(); ora.reset()
    torch.cuda.synchronize()
    log("  enqueue ...")
    t0 = time.perf_counter()
    eng.rollout_random(steps, obs, rew, done, reset_every=every, step0=0)
    log("  enqueued in %.1f us; synchronize ..." % ((time.perf_counter() - t0) * 1e6))
    torch.cuda.synchronize()
    log("  done after %.1f us" % ((time.perf_counter() - t0) * 1e6))
    last = {}
    for k in range(steps):
        if every and k % every == 0:
            ora.reset()
        _, o_obs, o_rew, _ = ora.step_random(want_obs=(k >= steps - ring))
        last[k] = (o_obs, o_rew)
    g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
    for k in range(max(0, steps - ring), steps):
        assert np.array_equal(g_rew[k % ring], last[k][1]), "rewards of step %d" % k
        assert np.array_equal(g_obs[k % ring], last[k][0]), "observations of step %d" % k
    a, b = eng.get_state(), ora.get_state()
    for key in ("world", "pos", "orient", "episode", "t"):
        assert np.array_equal(a[key], b[key]), key
    assert eng.status() == 0
    log("  bit-exact against the oracle")
    # a second call on the same buffers (cached argument set), interleaved with a HIP-launched step on the same stream
    eng.step_random(); ora.step_random()
    eng.rollout_random(5, obs, rew, done, reset_every=0, step0=steps)
    for k in range(5):
        _, o_obs, o_rew, _ = ora.step_random()
    torch.cuda.synchronize()
    k = steps + 4
    assert np.array_equal(rew.cpu().numpy()[k % ring], o_rew) and np.array_equal(obs.cpu().numpy()[k % ring], o_obs)
    log("  second call + interleaved HIP launch: bit-exact")
    eng.close()
log("aql smoke ok")
