#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer mode (SSD_HOST_PTRS) at 4096 envs: every call stages actions in and
observations / rewards / dones out and synchronises.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

E, N = 4096, 5
eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=N, seed=0)
eng.reset_host()
act = np.random.randint(0, 8, size=(E, N)).astype(np.int32)
for _ in range(20):
    eng.step_host(act)
t0 = time.perf_counter()
n = 200
for _ in range(n):
    eng.step_host(act)
dt = time.perf_counter() - t0
print("host-buffer mode, %d envs: %.1f us per step = %.1f M agent-env-steps/s (%.1f GB/s of observations over PCIe)"
      % (E, dt * 1e6 / n, E * N * n / dt / 1e6, E * N * 675 * n / dt / 1e9))
