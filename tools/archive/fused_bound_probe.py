#!/usr/bin/env python3
"""What bounds the fused rollout kernel?  Diagnostic library (make stamps): time it as is, with the observation stores
folded onto 64 envs' blocks (no HBM write stream), and without observations at all.  GPU box."""
import ctypes as C
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SSD_LIB_PATH", os.path.join(REPO, "sequential_social_dilemma_games_amd", "libssd_hip_stamps.so"))
import torch
from sequential_social_dilemma_games_amd import _capi, constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

game = K.GAME_CLEANUP if (len(sys.argv) > 1 and sys.argv[1] == "cleanup") else K.GAME_HARVEST
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = VecEngine(game, None, num_envs=E, num_agents=5, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
L = _capi.lib()
L.ssd_debug_set_skip.argtypes = [C.c_void_p, C.c_uint32]


def t(label, obs=True):
    r = ring if obs else (None, ring[1], ring[2])
    def go(n):
        _capi.check(L.ssd_rollout_random(eng._h, eng.num_actions, n, 1000, 0, eng._dp(r[0]) if obs else None, eng._dp(r[1]), eng._dp(r[2]),
                                         1, _capi.SSD_ROLLOUT_FUSED, eng._stream()), eng._h)
    go(300); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); go(3000); b.record(); torch.cuda.synchronize()
    print("%-70s %.2f us/step" % (label, a.elapsed_time(b) * 1e3 / 3000))


t("fused rollout")
L.ssd_debug_set_skip(eng._h, 16)
t("fused rollout, obs stores folded onto 64 envs' blocks (no write stream)")
L.ssd_debug_set_skip(eng._h, 32)
t("fused rollout, overlay phase skipped (wrong pixels; prices the overlay)")
L.ssd_debug_set_skip(eng._h, 0)
t("fused rollout, no observation output", obs=False)
