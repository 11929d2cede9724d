"""Pins the C oracle (oracle/ssd_oracle.c) to the reference: replays every transition and
reset that tests/golden/gen_golden.py recorded from the reference itself and demands
bit-exact world grid, beam overlay, positions, orientations, rewards and uint8 observations."""
import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle


@pytest.mark.parametrize("name", G.group_names())
def test_oracle_replays_reference_transitions(name):
    g = G.load(name)
    o = pyoracle.Oracle(g.game, g.map, 1, g.N, G.default_lut(), view_len=g.view_len, seed=g.seed, env_base=g.env)
    s = g.steps
    for k in range(g.n_steps):
        o.set_state(world=s["pre_world"][k][None], beam=np.zeros_like(s["pre_world"][k][None]),
                    pos=s["pre_pos"][k][None], orient=s["pre_orient"][k][None],
                    episode=np.array([s["episode"][k]], np.uint32), t=np.array([s["t"][k] - 1], np.uint32))
        obs, rew, done = o.step(s["act"][k][None], order=s["order"][k][None])
        st = o.get_state()
        where = "%s step %d" % (name, k)
        np.testing.assert_array_equal(st["pos"][0], s["pos"][k], err_msg=where)
        np.testing.assert_array_equal(st["orient"][0], s["orient"][k], err_msg=where)
        np.testing.assert_array_equal(st["world"][0], s["world"][k], err_msg=where)
        np.testing.assert_array_equal(st["beam"][0], s["beam"][k], err_msg=where)
        np.testing.assert_array_equal(rew[0], s["rew"][k], err_msg=where)
        np.testing.assert_array_equal(obs[0], s["obs"][k], err_msg=where)
        assert not done.any()
    r = g.resets
    for k in range(g.n_resets):
        ep = int(r["episode"][k])
        o.set_state(episode=np.array([(ep - 1) & 0xFFFFFFFF], np.uint32))
        obs = o.reset()
        st = o.get_state()
        where = "%s reset %d" % (name, k)
        assert st["episode"][0] == ep and st["t"][0] == 0
        np.testing.assert_array_equal(st["pos"][0], r["pos"][k], err_msg=where)
        np.testing.assert_array_equal(st["orient"][0], r["orient"][k], err_msg=where)
        np.testing.assert_array_equal(st["world"][0], r["world"][k], err_msg=where)
        np.testing.assert_array_equal(obs[0], r["obs"][k], err_msg=where)


def test_free_running_rollout_matches_reference():
    """No state injection: reset once, then feed the recorded actions; the oracle must track
    the reference for the whole Harvest rollout (first half, up to the mid-rollout reset)."""
    g = [x for x in G.groups() if x.name.endswith("harvest_16x38_n5_v7")][0]
    o = pyoracle.Oracle(g.game, g.map, 1, g.N, G.default_lut(), view_len=g.view_len, seed=g.seed, env_base=g.env)
    o.reset()
    s = g.steps
    k = 0
    while k < g.n_steps and s["episode"][k] == 0:
        obs, rew, _ = o.step(s["act"][k][None], order=s["order"][k][None])
        st = o.get_state()
        np.testing.assert_array_equal(st["world"][0], s["world"][k])
        np.testing.assert_array_equal(st["pos"][0], s["pos"][k])
        np.testing.assert_array_equal(obs[0], s["obs"][k])
        np.testing.assert_array_equal(rew[0], s["rew"][k])
        k += 1
    assert k > 100
