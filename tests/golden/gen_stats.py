#!/usr/bin/env python3
"""Statistics of the UNPATCHED reference (its own NumPy / random Mersenne Twisters) for the stochastic
parts of the path, written to tests/golden/stats_reference_rng.json.  Build container only.

Under the shared counter PRNG the engine is bit-exact (gen_golden.py); against the reference's native RNG
the claim is distributional.  Scenarios (all without state injection beyond the initial grid):
  harvest_regrowth : HarvestEnv(num_agents=0) on the default map with every second apple removed;
                     number of apples after 50 / 150 steps                  (harvest.py:75-104)
  cleanup_spawn    : CleanupEnv(num_agents=0) on the default map with the waste reduced to 20 cells;
                     apples and waste cells after 40 / 120 steps            (cleanup.py:132-171)
  harvest_rollout  : HarvestEnv(5 agents), uniform random actions, 200 steps: apples left, reward sum per env,
                     fraction of (agent, step) pairs hit by a beam          (whole step)
Each entry stores mean, standard deviation and the number of runs.
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402  (shims + reference import path; the proxies are NOT installed here)


def main():
    GG.install_shims()
    from social_dilemmas.envs.harvest import HarvestEnv
    from social_dilemmas.envs.cleanup import CleanupEnv
    out = {}

    def stat(xs):
        xs = np.asarray(xs, dtype=np.float64)
        return {"mean": float(xs.mean()), "std": float(xs.std(ddof=1)), "n": int(len(xs))}

    a50, a150 = [], []
    for s in range(150):
        np.random.seed(s); random.seed(s)
        env = HarvestEnv(num_agents=0)
        env.reset()
        for k, (r, c) in enumerate(env.apple_points):
            if k % 2:
                env.world_map[r, c] = ' '
        for t in range(150):
            env.step({})
            if t == 49:
                a50.append(int((env.world_map == 'A').sum()))
        a150.append(int((env.world_map == 'A').sum()))
    out["harvest_regrowth"] = {"apples_t50": stat(a50), "apples_t150": stat(a150)}
    print("harvest_regrowth", out["harvest_regrowth"])

    ap40, ap120, h40, h120 = [], [], [], []
    for s in range(150):
        np.random.seed(1000 + s); random.seed(1000 + s)
        env = CleanupEnv(num_agents=0)
        env.reset()
        hs = [p for p in env.waste_points if env.world_map[p[0], p[1]] == 'H']
        for k, (r, c) in enumerate(hs):
            if k >= 20:
                env.world_map[r, c] = 'R'
        for t in range(120):
            env.step({})
            if t == 39:
                ap40.append(int((env.world_map == 'A').sum())); h40.append(int((env.world_map == 'H').sum()))
        ap120.append(int((env.world_map == 'A').sum())); h120.append(int((env.world_map == 'H').sum()))
    out["cleanup_spawn"] = {"apples_t40": stat(ap40), "apples_t120": stat(ap120), "waste_t40": stat(h40), "waste_t120": stat(h120)}
    print("cleanup_spawn", out["cleanup_spawn"])

    left, rsum, hits = [], [], []
    for s in range(120):
        np.random.seed(2000 + s); random.seed(2000 + s)
        env = HarvestEnv(num_agents=5)
        env.reset()
        tot, nh = 0, 0
        for t in range(200):
            act = np.random.randint(8, size=5)
            _, rew, _, _ = env.step({'agent-%d' % i: int(act[i]) for i in range(5)})
            tot += sum(rew.values())
            nh += sum(1 for v in rew.values() if v <= -49)
        left.append(int((env.world_map == 'A').sum())); rsum.append(tot); hits.append(nh / 1000.0)
    out["harvest_rollout"] = {"apples_left_t200": stat(left), "reward_sum": stat(rsum), "hit_fraction": stat(hits)}
    print("harvest_rollout", out["harvest_rollout"])
    json.dump(out, open(os.path.join(HERE, "stats_reference_rng.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
