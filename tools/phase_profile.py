#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused step kernel (diagnostic build with -DSSD_STAMPS).

    make -C sequential_social_dilemma_games_amd/csrc stamps
    SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_stamps.so python tools/phase_profile.py [harvest|cleanup] [E]

Read the SHARES, not the totals: stamps cost a store each and pin the schedule."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SSD_LIB_PATH", os.path.join(REPO, "sequential_social_dilemma_games_amd", "libssd_hip_stamps.so"))

import torch  # noqa: E402
from sequential_social_dilemma_games_amd import _capi, constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402

NAMES = ["load state", "actions+move", "consume+occ", "beams", "respawn", "write-back", "overlay", "wg barrier", "obs"]


def main():
    game = K.GAME_CLEANUP if (len(sys.argv) > 1 and sys.argv[1] == "cleanup") else K.GAME_HARVEST
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    eng = VecEngine(game, None, num_envs=E, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    eng.reset(obs=out[0])
    for _ in range(200):
        eng.step_random(out=out)
    stamps = torch.zeros((E, 16), dtype=torch.int64, device="cuda")
    L = _capi.lib()
    L.ssd_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    L.ssd_debug_set_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
    acc = torch.zeros(9, dtype=torch.float64)
    span = 0.0
    sq = torch.zeros(5, dtype=torch.float64)
    eq = torch.zeros(5, dtype=torch.float64)
    reps = 50
    for _ in range(reps):
        eng.step_random(out=out)
        torch.cuda.synchronize()
        s = stamps.cpu().double()
        d = s[:, 1:10] - s[:, 0:9]
        acc += d.mean(dim=0)
        t0 = s[:, 10].min()                        # s_memrealtime: 100 MHz, common to all XCDs
        span += float(s[:, 11].max() - t0) * 0.01
        starts = (s[:, 10] - t0) * 0.01
        ends = (s[:, 11] - t0) * 0.01
        q = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
        sq += torch.quantile(starts, q)
        eq += torch.quantile(ends, q)
    acc /= reps
    tot = float(acc.sum())
    print("phase shares per wave (cycles of s_memtime; E=%d, %s)" % (E, "cleanup" if game else "harvest"))
    for n, v in zip(NAMES, acc.tolist()):
        print("  %-14s %9.0f  %5.1f %%" % (n, v, 100.0 * v / tot))
    print("  %-14s %9.0f" % ("sum", tot))
    print("  first stamp -> last stamp over the grid: %.2f us" % (span / reps))
    epb = int(os.environ.get("SSD_ENVS_PER_BLOCK", "0")) or None
    if epb:                                        # blocks go round-robin to the 8 XCDs
        xcd = (torch.arange(E) // epb) % 8
        for x in range(8):
            m = xcd == x
            print("  XCD %d: starts %.2f..%.2f us, ends %.2f..%.2f us (last launch)" % (
                x, float(starts[m].min()), float(starts[m].max()), float(ends[m].min()), float(ends[m].max())))
    print("  wave start times us (min / 10 %% / median / 90 %% / max): " + " / ".join("%.2f" % v for v in (sq / reps).tolist()))
    print("  wave end times   us (min / 10 %% / median / 90 %% / max): " + " / ".join("%.2f" % v for v in (eq / reps).tolist()))


if __name__ == "__main__":
    main()
