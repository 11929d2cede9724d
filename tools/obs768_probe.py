#!/usr/bin/env python3
"""Experiment (not the output layout): agent observation blocks 768 bytes apart, so that every store instruction of the renderer
covers whole 128-byte lines -- does the partly written line at each block's end cost anything once the output ring no longer
fits the memory-side cache?  Needs the experiment build (make exp EXP=-DSSD_EXP_OBS768); with any other library it measures the
ordinary layout in a buffer of the same size.   python tools/obs768_probe.py [ring] [steps]"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from _label import label  # noqa: E402
import torch  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402


def main():
    label("obs768_probe " + " ".join(sys.argv[1:]))
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    E, N = 4096, 5
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=N, seed=0)
    STRIDE = int(os.environ.get("OBS_STRIDE", "768"))
    obs = torch.empty((R, E, N, STRIDE), dtype=torch.uint8, device="cuda")          # room for either layout
    rew = torch.empty((R, E, N), dtype=torch.int32, device="cuda")
    done = torch.empty((R, E, N), dtype=torch.uint8, device="cuda")
    L = eng._L

    def call(n, step0):
        rc = L.ssd_rollout_random(eng._h, 8, n, 1000, step0, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), R, 0,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    call(128, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call(steps, 128)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / steps
    print("ring %d: %.3f us per step; path %s" % (R, us, eng.rollout_path()))


if __name__ == "__main__":
    main()
