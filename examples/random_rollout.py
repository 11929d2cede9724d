#!/usr/bin/env python3
"""Random-action rollouts in the style of the reference's rollout.py:48-82 (without the video writer),
once through the drop-in dict API and once through the batched tensor API.

    python examples/random_rollout.py [harvest|cleanup] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from sequential_social_dilemma_games_amd import CleanupEnv, HarvestEnv, VecEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "harvest"
    horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cls, game, action_dim = (HarvestEnv, 0, 8) if name == "harvest" else (CleanupEnv, 1, 9)

    # --- the reference's loop: one env object, dict in / dict out (rollout.py:62-70)
    env = cls(num_agents=5)
    env.reset()
    t0 = time.perf_counter()
    total = 0
    for _ in range(horizon):
        rand_action = np.random.randint(action_dim, size=5)
        obs, rew, dones, info = env.step({'agent-%d' % i: int(rand_action[i]) for i in range(5)})
        total += sum(rew.values())
    dt = time.perf_counter() - t0
    print("dict API   : %7.0f env-steps/s (%d steps, reward sum %d); the reference does ~458/s per CPU core"
          % (horizon / dt, horizon, total))
    frame = env.map_to_colors()                  # full-frame RGB, rendered on the GPU (rollout.py:77)
    print("             full frame %s, obs %s %s" % (frame.shape, obs['agent-0'].shape, obs['agent-0'].dtype))

    # --- the batched API: 4096 envs per kernel launch, observations stay on the GPU
    import torch
    eng = VecEngine(game, None, num_envs=4096, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    eng.reset(obs=out[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(horizon):
        eng.step_random(out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("batched API: %7.2f M agent-env-steps/s (4096 envs x 5 agents x %d steps, one Python call per step)"
          % (4096 * 5 * horizon / dt / 1e6, horizon))

    # --- the whole rollout as one library call (rollout.py:58-70): step launches enqueued from C, then one fused launch
    ring = tuple(t.unsqueeze(0) for t in out)     # 1 output slot: every step overwrites the same buffers
    for fused in (False, True):
        eng.rollout_random(horizon, *ring, reset_every=horizon, fused=fused)      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_random(horizon, *ring, reset_every=horizon, fused=fused)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("rollout API: %7.2f M agent-env-steps/s (%s)" % (4096 * 5 * horizon / dt / 1e6,
              "ONE kernel launch for the whole rollout" if fused else "one kernel launch per step and env range"))


if __name__ == "__main__":
    main()
