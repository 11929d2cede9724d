#!/usr/bin/env python3
"""Where the fixed cost of a short call sits on the time axis (diagnostic build, `make stamps`): the kernels' wave stamps
(s_memrealtime, 100 MHz) placed on the host's clock by a calibration kernel that publishes that counter to pinned memory.

    SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_stamps.so python tools/call_timeline.py [E]

Per call form: host call start -> first wave entry -> last env wave end -> last renderer wave end -> synchronize returns."""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SSD_LIB_PATH", os.path.join(REPO, "sequential_social_dilemma_games_amd", "libssd_hip_stamps.so"))

import torch  # noqa: E402
from sequential_social_dilemma_games_amd import _capi, constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402

now = lambda: time.clock_gettime_ns(time.CLOCK_MONOTONIC_RAW)


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    ring = tuple(t.unsqueeze(0) for t in out)
    eng.reset(obs=out[0])
    L = _capi.lib()
    L.ssd_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    L.ssd_debug_clock.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    stamps = torch.zeros((2 * E, 16), dtype=torch.int64, device="cuda")
    word = torch.zeros(8, dtype=torch.int64).pin_memory()
    arr = word.numpy()

    def calibrate():
        """offset (ns) such that host_ns = tick * 10 + offset: the freshest of many (host time, published tick) pairs"""
        arr[0] = 0
        assert L.ssd_debug_clock(eng._h, C.c_void_p(word.data_ptr()), 8000, None) == 0
        best = None
        t_end = now() + 800_000
        while now() < t_end:
            v = int(arr[0]); t = now()
            if v:
                d = t - v * 10
                best = d if best is None or d < best else best
        torch.cuda.synchronize()
        return best

    offs = [calibrate() for _ in range(5)]
    print("clock offset (ns), 5 calibrations: spread %.2f us" % ((max(offs) - min(offs)) / 1e3))
    print("  (the two clocks drift by ppm: %.2f us between the first and the last of them -- hence a calibration before and after every call)" % ((offs[-1] - offs[0]) / 1e3))
    for _ in range(50):
        eng.step_random(out=out)
    eng.rollout_random(20, *ring, reset_every=1000, step0=1)
    eng.rollout_random(1, *ring, reset_every=1000, step0=1)
    torch.cuda.synchronize()
    L.ssd_debug_set_stamps(eng._h, C.c_void_p(stamps.data_ptr()))

    def run(label, fn, reps=15, split=False):
        rows = []
        for rep in range(reps + 3):
            time.sleep(0.0005)
            stamps.zero_(); torch.cuda.synchronize()
            off0 = calibrate()
            time.sleep(0.0002)
            t0 = now(); fn(); t1 = now(); torch.cuda.synchronize(); t2 = now()
            off = (off0 + calibrate()) // 2
            to_host = lambda tick: tick * 10 + off
            s = stamps.cpu()
            a, b = s[:E], s[E:]
            ent, a_end = to_host(int(a[:, 14].min())), to_host(int(a[:, 11].max()))
            a_first = to_host(int(a[:, 10].min()))
            b_end = to_host(int(b[:, 11].max())) if split and int(b[:, 11].max()) > 0 else a_end
            if rep >= 3:
                rows.append(((t1 - t0) / 1e3, (t2 - t0) / 1e3, (ent - t0) / 1e3, (a_first - t0) / 1e3, (a_end - t0) / 1e3, (b_end - t0) / 1e3, (t2 - b_end) / 1e3))
        c = [med([r[i] for r in rows]) for i in range(7)]
        print("%-34s call returns %.1f | sync returns %.1f | LAST launch: first wave entry %.1f, first stamp %.1f, env waves done %.1f, renderers done %.1f | tail %.1f us"
              % (label, *c))

    run("ssd_step (HIP launch), 1 step", lambda: eng.step_random(out=out))
    for chains in (1, 2):
        eng.set_rollout_chains(chains)
        for n in (1, 2, 20):
            run("rollout chains=%d n=%d" % (chains, n), lambda: eng.rollout_random(n, *ring, reset_every=1000, step0=1), split=n >= 4)
    print("(the first-wave columns are those of the call's LAST step launch; with n=1 that is the only one)")


if __name__ == "__main__":
    main()
