#!/usr/bin/env python3
"""Run the step kernel with a phase-skip mask (diagnostic library, make stamps) -- the workload of tools/pmc_phases.sh,
which wraps it in rocprofv3 --pmc passes to get the DYNAMIC instruction counts of each phase as differences.

    python3 tools/pmc_phases.py <skip_mask> <obs 0|1> [harvest|cleanup] [E] [steps]
"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SSD_LIB_PATH", os.path.join(REPO, "sequential_social_dilemma_games_amd", "libssd_hip_stamps.so"))

import torch  # noqa: E402
from sequential_social_dilemma_games_amd import _capi, constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402


def main():
    mask, with_obs = int(sys.argv[1], 0), int(sys.argv[2])
    game = K.GAME_CLEANUP if (len(sys.argv) > 3 and sys.argv[3] == "cleanup") else K.GAME_HARVEST
    E = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 200
    eng = VecEngine(game, None, num_envs=E, num_agents=5, seed=0)
    obs, rew, done = eng.alloc_outputs()
    eng.reset(obs=obs)
    L = _capi.lib()
    L.ssd_debug_set_skip.argtypes = [C.c_void_p, C.c_uint32]
    L.ssd_debug_set_skip(eng._h, mask)
    dp, st = eng._dp, eng._stream()
    for _ in range(steps):
        L.ssd_step_random(eng._h, eng.num_actions, None, dp(obs) if with_obs else None, dp(rew), dp(done), 0, st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
