#!/usr/bin/env python3
"""What a rollout call costs when the caller's stream has work pending (the usual case inside a training loop): K-step calls
from an idle stream against calls that find a short kernel pending and must fork from the stream.  us per call, median of many.
With the test-hook build SSD_AQL_FORK_KIND=0 / 1 picks the fork: a barrier-AND packet on an HSA signal / a waiting kernel.
    python tools/fork_ab.py [K] [envs]"""
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
    label("fork_ab " + " ".join(sys.argv[1:]))
    Ks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    ring = tuple(t.unsqueeze(0) for t in out)
    x = torch.zeros(1 << 20, device="cuda")
    big = torch.zeros(1 << 26, device="cuda")
    x.add_(1.0); big.add_(1.0)
    for i in range(5):
        eng.rollout_random(Ks, *ring, reset_every=1000, step0=i * Ks)
    torch.cuda.synchronize()

    def run(pre, reps=41):
        vals = []
        for r in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if pre is not None:
                pre()
            eng.rollout_random(Ks, *ring, reset_every=1000, step0=100 + r * Ks)
            torch.cuda.synchronize()
            vals.append((time.perf_counter() - t0) * 1e6)
        return float(np.median(vals)), float(np.min(vals))
    idle = run(None)
    only_small = run(lambda: None)  # same as idle, ordering check
    small = run(lambda: x.add_(1.0))
    bigk = run(lambda: big.add_(1.0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        big.add_(1.0)
        torch.cuda.synchronize()
    big_us = (time.perf_counter() - t0) * 1e6 / 50
    p = eng.rollout_path()
    print("K=%d E=%d: idle stream %.1f us per call (min %.1f) | 4 MB kernel pending %.1f (min %.1f) | 256 MB kernel pending %.1f (min %.1f; that kernel + sync alone: %.1f) | forked %s"
          % (Ks, E, idle[0], idle[1], small[0], small[1], bigk[0], bigk[1], big_us, p["forked"]))


if __name__ == "__main__":
    main()
