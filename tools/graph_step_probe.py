#!/usr/bin/env python3
"""Can the per-call step be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed?  K ssd_step launches with the caller's
action tensors captured once, replayed with fresh actions written into the same tensors; every replayed step against the oracle;
us per step of a replay.   python tools/graph_step_probe.py [K]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch
import golden_util as G
from _label import label
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

label("graph_step_probe " + " ".join(sys.argv[1:]))
KS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E, N = 4096, 5
eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=N, seed=11)
ora = pyoracle.Oracle(K.GAME_HARVEST, K.HARVEST_MAP, E, N, G.default_lut(), seed=11)
eng.reset(); ora.reset()
acts = torch.zeros((KS, E, N), dtype=torch.int32, device="cuda")
outs = [eng.alloc_outputs() for _ in range(KS)]
rng = np.random.RandomState(5)
# warm-up on a side stream (as torch's capture recipe asks), then capture K steps
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    a0 = rng.randint(0, 8, size=(KS, E, N)).astype(np.int32); acts.copy_(torch.from_numpy(a0))
    for k in range(KS):
        eng.step(acts[k], out=outs[k])
        ora.step(a0[k])
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
a1 = rng.randint(0, 8, size=(KS, E, N)).astype(np.int32); acts.copy_(torch.from_numpy(a1))
views = [acts[k] for k in range(KS)]
with torch.cuda.graph(g):
    for k in range(KS):
        eng.step(views[k], out=outs[k])
torch.cuda.synchronize()
# (capture does not execute: the state is still the one after the warm-up)
ok = True
for rep in range(3):
    a = rng.randint(0, 8, size=(KS, E, N)).astype(np.int32); acts.copy_(torch.from_numpy(a))
    g.replay(); torch.cuda.synchronize()
    for k in range(KS):
        o_obs, o_rew, _ = ora.step(a[k])
        if not (np.array_equal(outs[k][0].cpu().numpy(), o_obs) and np.array_equal(outs[k][1].cpu().numpy(), o_rew)):
            ok = False; print("replay %d step %d differs from the oracle" % (rep, k)); break
print("graph replay bit-exact:", ok)
xs = []
for rep in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6)
xs.sort()
print("graph of %d per-call steps: %.1f us per replay = %.2f us per step (median of 20)" % (KS, xs[10], xs[10] / KS))
ys = []
for rep in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(KS):
        eng.step(views[k], out=outs[k])
    torch.cuda.synchronize(); ys.append((time.perf_counter() - t0) * 1e6)
ys.sort()
print("the same %d steps as Python calls: %.1f us = %.2f us per step" % (KS, ys[10], ys[10] / KS))
