"""The reference's own recorded rollouts replayed FREE-RUNNING through the fast dispatch paths.

tests/test_hip_parity.py injects every recorded pre-state and checks one transition at a time through ssd_step.  Here nothing is
injected: a fresh engine is reset and fed the recorded ACTION SEQUENCES -- g19-g25 (random rollouts on every map, Cleanup ones
biased towards CLEAN) and the return_agent_actions fixtures x0-x4 (x2 / x3: random subsets and action-dict orders, 12 agents) --
with ssd_rollout_actions, in calls of 1 .. 34 steps, (i) as chains of per-step launches in the library's own queues (coherent
kernels, split rendering from 4 steps on; the general kernels behind fences where an explicit action order or an agent count
has no map-specific kernel) and (ii) as the fused rollout kernel.  After every call: every step's observations and rewards of
the recorded env against the fixture's, and the env's world / positions / orientations against the fixture's post-state of the
call's last step.  The fixtures come from the imported reference (tests/golden/gen_golden*.py): this pins those kernels to the
reference directly, not via the oracle."""
import os

import numpy as np
import pytest

import golden_util as G
from sequential_social_dilemma_games_amd import constants as K  # noqa: F401
from sequential_social_dilemma_games_amd.engine import VecEngine

pytestmark = pytest.mark.gpu

NAMES = ["g19_harvest_16x38_n5_v7", "g20_harvest_16x38_n5_v7", "g21_harvest_16x38_n9_v7", "g22_cleanup_25x18_n5_v7",
         "g23_cleanup_25x18_n10_v7", "g24_harvest_25x38_n5_v7", "g25_cleanup_48x36_n10_v7",
         "extras/x0_harvest_16x38_n5_raa", "extras/x1_cleanup_25x18_n5_raa", "extras/x2_harvest_16x38_n5_raa_subsets",
         "extras/x3_harvest_16x38_n12_raa", "extras/x4_cleanup_25x18_n10_raa"]
CHUNKS = (1, 2, 3, 5, 8, 13, 21, 34)


def _index_order(g):
    for k in range(g.n_steps):
        present = [i for i in range(g.N) if g.steps["act"][k, i] >= 0]
        if [int(x) for x in g.steps["order"][k] if x != 255] != present:
            return False
    return True


@pytest.mark.parametrize("mode", ["chains", "fused"])
@pytest.mark.parametrize("name", NAMES)
def test_reference_rollouts_replayed_free_running(name, mode):
    import torch
    g = G.Group(os.path.join(G.GOLDEN_DIR, name + ".npz"))
    st = g.steps
    assert g.view_len == 7 and g.n_steps >= 20
    n, N = g.n_steps, g.N
    ep = st["episode"]
    k_reset = next(k for k in range(1, n) if ep[k] != ep[k - 1])       # the reference was reset once mid-way (gen_golden.py)
    assert 2 * k_reset > n - 1 and int(st["t"][0]) == 1 and int(ep[0]) == 0
    # the recorded env among 2560: in the second chain's range when its global index allows (env_index_base >= 0)
    E = 2560
    j = min(int(g.env), 1500)
    eng = VecEngine(g.game, g.map, num_envs=E, num_agents=N, seed=g.seed, env_index_base=int(g.env) - j)
    eng.set_rollout_chains(2 if mode == "chains" else 1)
    acts = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(st["act"][:, None, :], (n, E, N)))).cuda()
    order = None
    if not _index_order(g):
        order = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(st["order"][:, None, :], (n, E, N)))).cuda()
    R = 40
    obs = torch.zeros((R, E, N, 15, 15, 3), dtype=torch.uint8, device="cuda")
    rew = torch.zeros((R, E, N), dtype=torch.int32, device="cuda")
    done = torch.ones((R, E, N), dtype=torch.uint8, device="cuda")
    # (what is compared first is the DATA, call by call; how every call was dispatched is collected and asserted at the end, so
    # that a box on which the library's own queues are not available -- SSD_AQL=0, an unmatched HSA agent, a queue that fails its
    # probe -- still checks the parity of ssd_rollout_actions against the reference and reports the dispatch mismatch separately)
    k, call, seen, paths = 0, 0, set(), []
    while k < n:
        m = min(CHUNKS[call % len(CHUNKS)], n - k)
        eng.rollout_actions(acts, m, obs, rew, done, reset_every=k_reset, step0=k, fused=(mode == "fused"), order=order)
        path = eng.rollout_path()
        paths.append(path)
        seen.add((path["coherent"], path["split"]))
        torch.cuda.synchronize()
        got_obs = obs[:, j].cpu().numpy()
        got_rew = rew[:, j].cpu().numpy()
        for s in range(k, k + m):
            np.testing.assert_array_equal(got_rew[s % R], st["rew"][s], err_msg="%s: rewards of step %d" % (name, s))
            assert np.array_equal(got_obs[s % R], st["obs"][s]), "%s: observations of step %d differ from the reference's" % (name, s)
        a = eng.get_state()
        last = k + m - 1
        np.testing.assert_array_equal(a["world"][j], st["world"][last], err_msg="%s: world after step %d" % (name, last))
        np.testing.assert_array_equal(a["pos"][j], st["pos"][last], err_msg="%s: positions after step %d" % (name, last))
        np.testing.assert_array_equal(a["orient"][j], st["orient"][last], err_msg="%s: orientations after step %d" % (name, last))
        assert int(a["t"][j]) == int(st["t"][last]) and int(a["episode"][j]) == int(ep[last])
        k += m
        call += 1
    assert not done[:min(n, R)].any().item() and eng.status() == 0     # (slots no step wrote keep their initial 1)
    # ... and only now the dispatch path (every comparison above has passed: the data is right whatever path ran)
    for path in paths:
        if mode == "fused":
            assert path["fused"] and not path["aql"], "data matched the reference, but the call was not dispatched as the fused kernel: %r" % (path,)
        else:
            assert path["aql"] and path["chains"] == 2, ("data matched the reference, but the call did not go through the library's own "
                                                         "queues as 2 chains (SSD_AQL=0? agent match / queue probe fell back?): %r" % (path,))
    if mode == "chains":
        fast = order is None and N in (5, 10) and os.environ.get("SSD_AQL_COHERENT", "1") != "0"
        if fast:      # the map-specific kernels: coherent chains, split from 4 steps on -- both forms were exercised
            assert (True, False) in seen and ((True, True) in seen or os.environ.get("SSD_AQL_SPLIT", "1") == "0"), seen
        else:         # 9 / 12 agents, explicit orders: the general kernels behind agent-scope fences, still the library's queues
            assert seen == {(False, False)}, seen
    eng.close()
