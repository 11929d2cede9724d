#!/usr/bin/env python3
"""What the dispatch-queue probe (csrc/ssd_aql.hip, probe_pool_queue) sees, for calibration: run with SSD_AQL_VERBOSE=1 and
SSD_AQL_QUEUES=1..3, optionally with N torch side streams kept busy (argv[1]) and the probe switched off (test-hook build,
SSD_AQL_PROBE=0) to see what the process's launches cost with the queues kept.  Prints the plain-launch cost of the process before
and after the library's first rollout, and the path.   python tools/queue_probe.py [side streams] [chains]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from _label import label  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402


def main():
    label("queue_probe " + " ".join(sys.argv[1:]))
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    chains = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    side = [torch.cuda.Stream() for _ in range(n_side)]
    bufs = [torch.zeros(1 << 22, device="cuda") for _ in side]

    def keep_busy(n):
        for s, t in zip(side, bufs):
            with torch.cuda.stream(s):
                for _ in range(n):
                    t.add_(1.0)
    y = torch.zeros(256, device="cuda")

    def launch_cost():
        vals = []
        for _ in range(7):
            keep_busy(50)
            torch.cuda.synchronize()
            keep_busy(400)
            t0 = time.perf_counter()
            for _ in range(200):
                y.add_(1.0)
            torch.cuda.current_stream().synchronize()
            vals.append((time.perf_counter() - t0) * 1e6 / 200)
            torch.cuda.synchronize()
        return float(np.median(vals))
    keep_busy(100)
    torch.cuda.synchronize()
    before = launch_cost()
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=6144, num_agents=5, seed=3)
    eng.set_rollout_chains(chains)
    out = eng.alloc_outputs()
    ring = tuple(t.unsqueeze(0) for t in out)
    keep_busy(300)
    eng.rollout_random(12, *ring, reset_every=1000, step0=0)
    torch.cuda.synchronize()
    path = eng.rollout_path()
    after = launch_cost()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.rollout_random(200, *ring, reset_every=1000, step0=12)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / 200
    print("side streams %d: plain launch %.2f us before, %.2f us after the first rollout; rollout %.2f us per 6144-env step; path %s"
          % (n_side, before, after, us, path))


if __name__ == "__main__":
    main()
