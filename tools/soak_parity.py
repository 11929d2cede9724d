#!/usr/bin/env python3
"""Long-run parity soak (GPU box): E envs, T random-action steps with a reset every 1000 steps; every K steps the
engine's full state (world, positions, orientations, counters) and the step's observations / rewards are compared
with the C oracle's.  python tools/soak_parity.py [harvest|cleanup|harvest25x38|cleanup48x36] [E] [T] [K] [step|chains|fused|actions|actions_fused]
(step: one step_random call per step; chains: ssd_rollout_random with 2 chains between checkpoints; fused: the rollout
kernel, one launch between checkpoints; actions / actions_fused: the same two with CALLER-SUPPLIED actions, ssd_rollout_actions,
a fresh random action tensor per chunk and the oracle stepped with the same actions).  The first line of the output says which
library ran under which SSD_* settings (tools/_label.py).  SOAK_EXPECT_PATH=sync|aql: the dispatch path the calls must report."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_util as G  # noqa: E402
from oracle import pyoracle  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402
sys.path.insert(0, os.path.join(REPO, "tools"))
from _label import label  # noqa: E402


def main():
    label("soak_parity " + " ".join(sys.argv[1:]))
    which = sys.argv[1] if len(sys.argv) > 1 else "harvest"   # harvest | cleanup | harvest25x38 | cleanup48x36 (the enlarged maps, 5 / 10 agents)
    game = K.GAME_CLEANUP if which.startswith("cleanup") else K.GAME_HARVEST
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    Kc = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    how = sys.argv[5] if len(sys.argv) > 5 else "step"
    amap = K.HARVEST_MAP if game == K.GAME_HARVEST else K.CLEANUP_MAP
    n_agents = 5
    if which == "harvest25x38":
        amap = K.harvest_map_25x38()
    elif which == "cleanup48x36":
        amap, n_agents = K.cleanup_map_48x36(), 10
    eng = VecEngine(game, amap, num_envs=E, num_agents=n_agents, seed=2024)
    ora = pyoracle.Oracle(game, amap, E, n_agents, G.default_lut(), seed=2024)
    out = eng.alloc_outputs()
    ora.reset()
    if len(sys.argv) <= 5 or sys.argv[5] == "step":       # (the rollout calls reset at step 0 themselves: reset_every)
        eng.reset(obs=out[0])
    t0 = time.time()
    checks = 0
    rsum = 0
    R = int(os.environ.get("SOAK_RING", "1"))                  # output ring slots
    import torch
    ring = tuple(t.unsqueeze(0) for t in out) if R == 1 else tuple(torch.zeros((R,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in out)
    if how != "step":
        assert 1000 % Kc == 0
        eng.set_rollout_chains(2 if how in ("chains", "actions") else 1)
    use_actions = how.startswith("actions")
    na = 8 if game == K.GAME_HARVEST else 9
    rng = np.random.RandomState(99)
    a_dev = torch.zeros((Kc, E, n_agents), dtype=torch.int32, device="cuda") if use_actions else None
    expect = os.environ.get("SOAK_EXPECT_PATH")
    # SOAK_CHECK_ALL=1 (with SOAK_RING >= K): EVERY step's observations and rewards are compared, each in its ring slot, after the
    # call that wrote them -- what a ring beyond the memory-side cache needs (its stores take another path: write-back,
    # non-temporal, an agent-scope release once per round of the ring)
    check_all = os.environ.get("SOAK_CHECK_ALL", "0") == "1"
    assert not check_all or (how != "step" and R >= Kc)
    chunk_want = {}
    for s in range(T):
        if s and s % 1000 == 0:
            ora.reset()
            if how == "step":
                eng.reset(obs=out[0])
        want_obs = (s % Kc == Kc - 1) or check_all
        if how == "step":
            obs, rew, _ = eng.step_random(out=out)
        elif s % Kc == 0:                                      # the Kc steps up to the next checkpoint in one library call
            if use_actions:
                a_host = rng.randint(-1, na, size=(Kc, E, n_agents)).astype(np.int32)
                a_dev.copy_(torch.from_numpy(a_host))
                eng.rollout_actions(a_dev, Kc, *ring, reset_every=1000, step0=s, fused=how.endswith("fused"))
            else:
                eng.rollout_random(Kc, *ring, reset_every=1000, step0=s, fused=(how == "fused"))
            if expect:
                assert eng.rollout_path()[expect], eng.rollout_path()
            last = (s + Kc - 1) % R
            obs, rew = ring[0][last], ring[1][last]
        if use_actions:
            o_obs, o_rew, _ = ora.step(a_host[s % Kc])
        else:
            _, o_obs, o_rew, _ = ora.step_random(want_obs=want_obs)
        if check_all:
            chunk_want[s] = (o_obs, o_rew)
            if s % Kc == Kc - 1:
                g_obs, g_rew = ring[0].cpu().numpy(), ring[1].cpu().numpy()
                for k, (w_obs, w_rew) in chunk_want.items():
                    assert np.array_equal(g_rew[k % R], w_rew), "rewards of step %d (slot %d) differ after the call ending at step %d" % (k, k % R, s)
                    assert np.array_equal(g_obs[k % R], w_obs), "observations of step %d (slot %d) differ after the call ending at step %d" % (k, k % R, s)
                chunk_want.clear()
        if want_obs and s % Kc == Kc - 1:
            r = rew.cpu().numpy()
            assert np.array_equal(r, o_rew), "rewards differ at step %d" % s
            assert np.array_equal(obs.cpu().numpy(), o_obs), "observations differ at step %d" % s
            a, b = eng.get_state(), ora.get_state()
            for k in ("world", "pos", "orient", "episode", "t"):
                assert np.array_equal(a[k], b[k]), "%s differs at step %d" % (k, s)
            checks += 1
            rsum += int(r.sum())
            if checks % 10 == 0:
                print("step %6d ok (%d checkpoints, %.0f s)" % (s + 1, checks, time.time() - t0), flush=True)
    assert eng.status() == 0
    print("soak ok (%s): %s, %d envs x %d steps = %.1f M env-steps, %d checkpoints bit-exact" %
          (how, which, E, T, E * T / 1e6, checks))


if __name__ == "__main__":
    main()
