#!/usr/bin/env python3
"""Workload of tools/pmc_roles_phases.sh: rollout chains (split launches) of the diagnostic library with a phase-skip mask --
    python3 tools/pmc_roles_run.py <skip_mask> [harvest|cleanup|cleanup48x36] [E] [steps]
(the skipped phases make the results wrong; what is read is the instruction count of what is left)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SSD_LIB_PATH", os.path.join(REPO, "sequential_social_dilemma_games_amd", "libssd_hip_stamps.so"))
import torch  # noqa: E402
from sequential_social_dilemma_games_amd import _capi, constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402

mask = int(sys.argv[1], 0)
which = sys.argv[2] if len(sys.argv) > 2 else "harvest"
E = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
fused = os.environ.get("PMC_FUSED") == "1"            # the rollout kernel instead of the chains (tools/pmc_fused_phases.sh)
game = K.GAME_CLEANUP if which.startswith("cleanup") else K.GAME_HARVEST
amap, n = (K.cleanup_map_48x36(), 10) if which == "cleanup48x36" else (None, 5)
eng = VecEngine(game, amap, num_envs=E, num_agents=n, seed=0)
out = eng.alloc_outputs()
ring = tuple(t.unsqueeze(0) for t in out)
eng.reset(obs=out[0])
L = _capi.lib()
L.ssd_debug_set_skip.argtypes = [C.c_void_p, C.c_uint32]
eng.set_rollout_chains(1 if fused else 2)
eng.rollout_random(8, *ring, fused=fused)
torch.cuda.synchronize()
L.ssd_debug_set_skip(eng._h, mask)
for _ in range(3):
    eng.rollout_random(steps, *ring, fused=fused)
    torch.cuda.synchronize()
print("path", eng.rollout_path())
