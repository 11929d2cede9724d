#!/usr/bin/env python3
"""Launch-time of the kernel's modes at E envs (GPU box): step with / without observation output,
observe only, reset only.  Separates the per-launch floor from the phases."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402


def timeit(fn, n=2000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def time_rollout(eng, obs, rew, done, n=3000):
    """Launch-bound on the host otherwise: n steps enqueued by ONE library call (ssd_rollout_random)."""
    ro = None if obs is None else obs.unsqueeze(0)
    rr = None if rew is None else rew.unsqueeze(0)
    rd = None if done is None else done.unsqueeze(0)
    L, h, st, dp = eng._L, eng._h, eng._stream(), eng._dp

    def go(k):
        L.ssd_rollout_random(h, eng.num_actions, k, 0, 0, dp(ro), dp(rr), dp(rd), 1, 0, st)
    go(300)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    go(n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    game = K.GAME_CLEANUP if (len(sys.argv) > 2 and sys.argv[2] == "cleanup") else K.GAME_HARVEST
    eng = VecEngine(game, None, num_envs=E, num_agents=5, seed=0)
    obs, rew, done = eng.alloc_outputs()
    eng.reset(obs=obs)
    L, h, st = eng._L, eng._h, eng._stream()
    dp = eng._dp
    print("E=%d %s" % (E, "cleanup" if game else "harvest"))
    print("  (steps enqueued by ssd_rollout_random: device time, not host call rate)")
    print("  step_random, obs+rew+done : %6.2f us" % time_rollout(eng, obs, rew, done))
    print("  step_random, no obs       : %6.2f us" % time_rollout(eng, None, rew, done))
    print("  step_random, no outputs   : %6.2f us" % time_rollout(eng, None, None, None))
    print("  observe only              : %6.2f us" % timeit(lambda: L.ssd_observe(h, dp(obs), 0, st)))
    print("  reset, obs                : %6.2f us" % timeit(lambda: L.ssd_reset(h, None, dp(obs), 0, st)))
    print("  reset, no obs             : %6.2f us" % timeit(lambda: L.ssd_reset(h, None, None, 0, st)))
    if hasattr(L, "ssd_debug_set_skip") or os.environ.get("SSD_LIB_PATH", "").endswith("stamps.so"):
        L.ssd_debug_set_skip.argtypes = [C.c_void_p, C.c_uint32]
        full = time_rollout(eng, obs, rew, done)
        print("  marginal cost of a phase = full step (%.2f us) - step with the phase skipped:" % full)
        for bit, name in enumerate(("move", "consume+occupancy", "beams", "respawn")):
            L.ssd_debug_set_skip(h, 1 << bit)
            t = time_rollout(eng, obs, rew, done)
            print("    %-18s %5.2f us" % (name, full - t))
        L.ssd_debug_set_skip(h, 0xF)
        t = time_rollout(eng, None, rew, done)
        print("    all four + no obs: step costs %.2f us (load + write-back + launch floor)" % t)
        L.ssd_debug_set_skip(h, 16)
        t = time_rollout(eng, obs, rew, done)
        print("    obs stores folded onto 64 envs' blocks (same instructions, no HBM write stream): %.2f us" % t)
        L.ssd_debug_set_skip(h, 0)
    x = torch.zeros(1, device="cuda")
    print("  torch x.add_(1) (launch floor of a trivial kernel): %6.2f us" % timeit(lambda: x.add_(1)))


if __name__ == "__main__":
    main()
