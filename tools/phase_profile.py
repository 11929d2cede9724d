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
    which = sys.argv[1] if len(sys.argv) > 1 else "harvest"
    game = K.GAME_CLEANUP if which.startswith("cleanup") else K.GAME_HARVEST
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    amap, n_agents = {"cleanup48x36": (K.cleanup_map_48x36(), 10), "harvest25x38": (K.harvest_map_25x38(), 5)}.get(which, (None, 5))
    eng = VecEngine(game, amap, num_envs=E, num_agents=n_agents, seed=0)
    out = eng.alloc_outputs()
    eng.reset(obs=out[0])
    for _ in range(200):
        eng.step_random(out=out)
    stamps = torch.zeros((2 * E, 16), dtype=torch.int64, device="cuda")   # [E:] the renderer waves of split rollouts
    L = _capi.lib()
    L.ssd_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    L.ssd_debug_set_stamps(eng._h, C.c_void_p(stamps.data_ptr()))
    acc = torch.zeros(9, dtype=torch.float64)
    span = 0.0
    sq = torch.zeros(5, dtype=torch.float64)
    eq = torch.zeros(5, dtype=torch.float64)
    reps = 50
    burst = int(os.environ.get("SSD_PROFILE_BURST", "20"))     # launches back to back; the stamps are the last one's
    chains = int(os.environ.get("SSD_PROFILE_CHAINS", "0"))   # > 0: the burst goes through ssd_rollout_random with that many chains
    ring = tuple(t.unsqueeze(0) for t in out)
    if chains:
        eng.set_rollout_chains(chains)
    for _ in range(reps):
        if chains:
            eng.rollout_random(burst, *ring)
        else:
            for _ in range(burst):
                eng.step_random(out=out)
        torch.cuda.synchronize()
        s_all = stamps.cpu().double()
        s, sb = s_all[:E], s_all[E:]
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
    # the last launch: wave duration by what the env had to do (notes written by the diagnostic build)
    dur = (s[:, 11] - s[:, 10]) * 0.01
    cyc = s[:, 9] - s[:, 0]
    slow, shots = s[:, 12] > 0, s[:, 13]
    print("  last launch, wave duration (us, 100 MHz clock) / cycles by work:")
    for name, m in (("fast move, 0 shooters", (~slow) & (shots == 0)), ("fast move, 1 shooter", (~slow) & (shots == 1)),
                    ("fast move, 2+ shooters", (~slow) & (shots >= 2)), ("slow move, 0 shooters", slow & (shots == 0)),
                    ("slow move, 1+ shooters", slow & (shots >= 1))):
        if m.any():
            print("    %-24s %5.1f %% of envs: mean %.2f us, max %.2f us, mean %.0f cycles" % (
                name, 100.0 * float(m.double().mean()), float(dur[m].mean()), float(dur[m].max()), float(cyc[m].mean())))
    d_last = s[:, 1:10] - s[:, 0:9]
    for code, name in ((0, "move: fast path"), (1, "move: slow path, no contested cell"), (2, "move: slow path, contested cell (shuffle)")):
        m = s[:, 12] == code
        if m.any():
            print("    %-42s %5.2f %% of envs: move phase mean %.0f cycles, max %.0f" % (
                name, 100.0 * float(m.double().mean()), float(d_last[m][:, 1].mean()), float(d_last[m][:, 1].max())))
    for code, name in ((0, "beams: one parallel pass"), (1, "beams: more shooters than slots"), (2, "beams: conflict -> one by one")):
        m = (s[:, 15] == code) & (shots >= 1)
        if m.any():
            print("    %-32s %5.2f %% of envs: beams phase mean %.0f cycles, wave mean %.2f us, max %.2f us" % (
                name, 100.0 * float(m.double().mean()), float(d_last[m][:, 3].mean()), float(dur[m].mean()), float(dur[m].max())))
    idx = dur.sort().indices
    for name, sel in (("fastest 10 %", idx[: E // 10]), ("middle 10 %", idx[E // 2 - E // 20: E // 2 + E // 20]), ("slowest 1 %", idx[-max(E // 100, 1):])):
        print("    %-13s phases (cycles): " % name + " ".join("%s %.0f" % (n.split()[0], v) for n, v in zip(NAMES, d_last[sel].mean(dim=0).tolist()) if n != "wg barrier")
              + " | starts %.2f us" % float(starts[sel].mean()))
    if os.environ.get("SSD_SHOW_SLOWEST"):
        for w in idx[-int(os.environ["SSD_SHOW_SLOWEST"]):].tolist():
            print("      env %5d: %.2f us (start %.2f) slow-move %d shooters %d beams-code %d | " % (
                w, float(dur[w]), float(starts[w]), int(s[w, 12]), int(s[w, 13]), int(s[w, 15]))
                + " ".join("%s %d" % (n.split()[0], v) for n, v in zip(NAMES, d_last[w].tolist()) if n != "wg barrier"))
    if chains and float(sb[:, 10].max()) > 0:
        # split rollouts: the last launch's env waves (step k) and the renderer waves beside them / behind them.  The stamps of a
        # renderer wave are those of the LAST launch it ran in: the renderer-only launch that ends the call (after the last step).
        print("    dispatch path of the call: %s" % (eng.rollout_path(),))
        for c in range(chains):
            lo, hi = E * c // chains, E * (c + 1) // chains
            a0, a1 = s[lo:hi, 10], s[lo:hi, 11]
            b0, b1 = sb[lo:hi, 10], sb[lo:hi, 11]
            base = float(a0.min())
            us = lambda x: (float(x) - base) * 0.01
            print("    chain %d: last step's env waves start %.2f..%.2f end %.2f..%.2f us | renderer-only launch after it: waves start %.2f..%.2f end %.2f..%.2f us"
                  % (c, us(a0.min()), us(a0.max()), us(a1.min()), us(a1.max()), us(b0.min()), us(b0.max()), us(b1.min()), us(b1.max())))
        bd = sb[:, 1:4] - sb[:, 0:3]
        print("    renderer wave phases (cycles, mean): load snapshot %.0f, render + issue stores %.0f, stores landed %.0f; wave duration (us) median %.2f max %.2f"
              % (float(bd[:, 0].mean()), float(bd[:, 1].mean()), float(bd[:, 2].mean()), float(((sb[:, 11] - sb[:, 10]) * 0.01).median()),
                 float(((sb[:, 11] - sb[:, 10]) * 0.01).max())))
    if chains:
        for c in range(chains):
            lo, hi = E * c // chains, E * (c + 1) // chains
            print("    chain %d (envs %d..%d): waves start %.2f..%.2f us, end %.2f..%.2f us" % (
                c, lo, hi - 1, float(starts[lo:hi].min()), float(starts[lo:hi].max()), float(ends[lo:hi].min()), float(ends[lo:hi].max())))
    # position of the wave among the waves of its SIMD: envs e, e+? share a SIMD -- unknown mapping; report by end rank instead
    print("    wave entry -> first stamp (kernel-argument fetch): mean %.2f us, max %.2f us" % (
        float(((s[:, 10] - s[:, 14]) * 0.01).mean()), float(((s[:, 10] - s[:, 14]) * 0.01).max())))
    order = dur.sort().values
    print("    duration percentiles (us): " + " / ".join("%.2f" % float(order[int(q * (E - 1))]) for q in (0, 0.1, 0.5, 0.9, 0.99, 1.0)))
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
