"""Distributional agreement with the UNPATCHED reference (its own Mersenne Twisters): regrowth curves,
Cleanup spawn dynamics and random-rollout aggregates recorded by tests/golden/gen_stats.py, against the
oracle (CPU) and the HIP engine (GPU).  Tolerance: 4.5 standard errors of the difference of means."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import stats_scenarios as S
from oracle import pyoracle

REF = json.load(open(os.path.join(G.GOLDEN_DIR, "stats_reference_rng.json")))


class OracleBackend(object):
    def __init__(self, game, amap, E, N):
        self.o = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=4242)

    def reset(self):
        self.o.reset()

    def get_state(self):
        return self.o.get_state()

    def set_state(self, **kw):
        self.o.set_state(**kw)

    def step(self, act):
        self.o.step(act)

    def step_random(self):
        return self.o.step_random(want_obs=False)[2]


class HipBackend(object):
    def __init__(self, game, amap, E, N):
        from sequential_social_dilemma_games_amd.engine import VecEngine
        self.e = VecEngine(game, amap, num_envs=E, num_agents=N, seed=777)
        self.out = self.e.alloc_outputs()
        import torch
        self.noact = torch.zeros((E, N), dtype=torch.int32, device="cuda")

    def reset(self):
        self.e.reset(obs=self.out[0])

    def get_state(self):
        return self.e.get_state()

    def set_state(self, **kw):
        self.e.set_state(**kw)

    def step(self, act):
        self.e.step(self.noact, out=self.out)

    def step_random(self):
        return self.e.step_random(out=self.out)[1].cpu().numpy()


@pytest.mark.parametrize("name", sorted(S.SCENARIOS))
def test_oracle_statistics_match_the_reference_rng(name):
    S.check(REF[name], S.SCENARIOS[name](OracleBackend, 600), "oracle " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(S.SCENARIOS))
def test_engine_statistics_match_the_reference_rng(name):
    S.check(REF[name], S.SCENARIOS[name](HipBackend, 4096), "engine " + name)
