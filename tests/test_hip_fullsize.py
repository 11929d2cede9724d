"""Parity at BASELINE.json's full sizes (configs[1], configs[2]: 4096 envs x 5 agents on one GPU) and the
size-independent properties the domain offers: shard invariance (configs[3]'s partitioning), determinism,
wall / agent conservation, observation == re-render of the stored state."""
import os

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("game", [K.GAME_HARVEST, K.GAME_CLEANUP])
def test_full_size_rollout_is_bit_exact_against_the_oracle(game):
    """4096 envs x 5 agents, device-tensor API (what bench.py times), 30 random-action steps + a reset:
    uint8 observations, rewards and the full engine state equal the oracle's at every step."""
    import torch
    amap = K.HARVEST_MAP if game == K.GAME_HARVEST else K.CLEANUP_MAP
    E, N = 4096, 5
    eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=0)
    ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=0)
    out = eng.alloc_outputs()
    acts = torch.empty((E, N), dtype=torch.int32, device="cuda")
    np.testing.assert_array_equal(eng.reset(obs=out[0]).cpu().numpy(), ora.reset())
    for s in range(30):
        if s == 17:
            np.testing.assert_array_equal(eng.reset(obs=out[0]).cpu().numpy(), ora.reset())
        obs, rew, done = eng.step_random(out=out, actions_out=acts)
        o_act, o_obs, o_rew, _ = ora.step_random()
        np.testing.assert_array_equal(acts.cpu().numpy(), o_act, err_msg="actions, step %d" % s)
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew, err_msg="rewards, step %d" % s)
        assert np.array_equal(obs.cpu().numpy(), o_obs), "observations differ at step %d" % s
        assert not done.any()
    a, b = eng.get_state(), ora.get_state()
    for k in ("world", "pos", "orient", "episode", "t"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert eng.status() == 0


HOOKS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sequential_social_dilemma_games_amd",
                         "libssd_hip_testhooks.so")


@pytest.mark.parametrize("knobs", [dict(SSD_AQL_ALTERNATE="1", SSD_AQL_ALWAYS_FORK="1"),
                                   dict(SSD_AQL_ALTERNATE="1", SSD_AQL_SPLIT="0"),
                                   dict(SSD_AQL_COHERENT="0"), dict(SSD_AQL="0")])
@pytest.mark.parametrize("how", ["chains", "actions"])
@pytest.mark.parametrize("game", ["harvest", "cleanup"])
def test_rollout_dispatch_modes_when_envs_change_xcd(game, how, knobs):
    """The coherent chains of the rollout calls carry no cache write-back or invalidate between launches: an env's state --
    and, in split rollouts, the beam list its observations are rendered from -- must reach the next launch through memory
    wherever that launch's wave runs.  SSD_AQL_ALTERNATE halves the envs per workgroup on odd steps, so that most envs change
    workgroup, and with it XCD and L2, from every launch to the next (and the renderer workgroups of a launch read what the
    OTHER mapping wrote); SSD_AQL_ALWAYS_FORK makes every call fork from its stream.  4096 envs x 600 steps as rollout chains
    (device-drawn actions, and caller-supplied ones: ssd_rollout_actions), checked against the oracle every 100 steps.  Also:
    the same without split rendering, the plain kernels behind agent-scope fences, and the hipLaunchKernel path.  A process of
    its own each (the knobs are read once per process); the two test hooks exist in the test-hook build of the library only
    (libssd_hip_testhooks.so: the product sources with -DSSD_TESTHOOKS), which is what these processes load."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("SSD_AQL", "SSD_AQL_ALTERNATE", "SSD_AQL_ALWAYS_FORK", "SSD_AQL_SPLIT", "SSD_AQL_COHERENT"):
        env.pop(k, None)
    env.update(knobs)
    env["SSD_LIB_PATH"] = HOOKS_LIB
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_parity.py"), game, "4096", "600", "100", how],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "soak ok" in out, out[-2000:]


@pytest.mark.parametrize("how,hooks", [("chains", True), ("chains", False), ("actions", False), ("fused", False),
                                       ("chains", "SSD_AQL=0"), ("chains", "SSD_AQL_COHERENT=0")])
def test_output_rings_beyond_the_memory_side_cache(how, hooks):
    """An observation ring of more than 232 MB (here 21 slots x 4096 envs = 290 MB) is written with non-temporal write-BACK
    stores (ssd_kernels.hip select(): the partly covered sectors at the ends of the agents' 675-byte blocks merge in L2), made
    safe by an agent-scope release on one launch per round of the ring and the call's closing release.  100 steps in 20-step
    calls -- the ring goes round four times -- every step's observations and rewards compared with the oracle's in its slot
    after the call that wrote it.  With the test-hook library's SSD_AQL_ALTERNATE the env -> workgroup -> XCD mapping changes
    from every launch to the next, and 21 being odd, a slot's bytes are rewritten through the OTHER mapping a round later.
    Also through the hipLaunchKernel path (SSD_AQL=0) and with the plain kernels behind agent-scope fences (SSD_AQL_COHERENT=0),
    whose launches release after every step."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("SSD_AQL", "SSD_AQL_ALTERNATE", "SSD_AQL_ALWAYS_FORK", "SSD_AQL_SPLIT", "SSD_AQL_COHERENT", "SSD_LIB_PATH"):
        env.pop(k, None)
    if hooks is True:
        env.update(SSD_AQL_ALTERNATE="1", SSD_LIB_PATH=HOOKS_LIB)
    elif hooks:                                             # (a product knob: the hipLaunchKernel path / the plain kernels behind fences)
        k, v = hooks.split("=")
        env[k] = v
    env.update(SOAK_RING="21", SOAK_CHECK_ALL="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_parity.py"), "harvest", "4096", "100", "20", how],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "soak ok" in out, out[-2000:]


def test_join_wait_is_bounded():
    """the stream-side wait of a rollout's join (a one-wave kernel polling a counter the library's queues
    bump) must not be able to spin forever.  Test hook SSD_AQL_TEST_LOST_JOIN makes it wait for a count that never comes, with
    the time bound cut to 50 ms: the wave gives up, sets SSD_ST_WAIT_TIMEOUT and the stream goes on; the next call works."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sequential_social_dilemma_games_amd import _capi, constants as K\n"
        "from sequential_social_dilemma_games_amd.engine import VecEngine\n"
        "eng = VecEngine(K.GAME_HARVEST, None, num_envs=512, num_agents=5, seed=1)\n"
        "obs = torch.zeros((2, 512, 5, 15, 15, 3), dtype=torch.uint8, device='cuda')\n"
        "eng.reset()\n"
        "torch.cuda.synchronize()\n"
        "t0 = time.time()\n"
        "eng.rollout_random(6, obs, None, None, reset_every=0, step0=0)\n"
        "torch.cuda.synchronize()\n"
        "assert eng.rollout_path()['aql'] and not eng.rollout_path()['sync']\n"
        "st = eng.status()\n"
        "print('status', st, 'seconds', round(time.time() - t0, 3))\n"
        "assert st & _capi.SSD_ST_WAIT_TIMEOUT, st\n"
        "assert time.time() - t0 < 5.0\n"
        # the timeout is STICKY: the next rollout call drains the library's queues on the host, reports the\n"
        # condition once and takes the handle off that path; the call after it steps through hipLaunchKernel\n"
        "try:\n"
        "    eng.rollout_random(2, obs, None, None, reset_every=0, step0=6)\n"
        "    raise SystemExit('the call after a timed-out join did not report it')\n"
        "except _capi.SsdError as e:\n"
        "    assert 'timed out' in str(e), str(e)\n"
        "eng.rollout_random(2, obs, None, None, reset_every=0, step0=6)\n"
        "torch.cuda.synchronize()\n"
        "assert not eng.rollout_path()['aql'], eng.rollout_path()\n"
        "print('sticky ok')\n" % root)
    env = dict(os.environ, SSD_AQL_TEST_LOST_JOIN="1", SSD_AQL_TEST_TIMEOUT_MS="50", SSD_LIB_PATH=HOOKS_LIB)
    env.pop("SSD_AQL_SYNC", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "status" in out and "sticky ok" in out, out[-2000:]


def test_sync_mode_gives_the_same_results():
    """With a profiling tool attached (or SSD_AQL_SYNC=1) the rollout calls wait on the host instead of with kernels that wait
    for other queues' kernels (rocprofv3 --pmc runs kernels one at a time).  Same launches, same results; the path says so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSD_AQL_SYNC="1", SOAK_EXPECT_PATH="sync")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_parity.py"), "cleanup", "4096", "200", "50", "chains"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "soak ok" in out, out[-2000:]


def test_dispatch_queue_ring_wraps_many_times():
    """The library's dispatch queues are ring buffers it writes itself (ssd_aql.hip): with the test hook SSD_AQL_QUEUE_SIZE=64 a
    600-step rollout of 2 chains goes round each ring ~20 times -- packets, the every-fourth-step doorbells (never across the ring's
    end: next_slot), the wait for a slot's previous occupant -- against the oracle at every checkpoint."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSD_AQL_QUEUE_SIZE="64", SSD_LIB_PATH=HOOKS_LIB, SOAK_EXPECT_PATH="aql")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_parity.py"), "harvest", "2304", "600", "50", "chains"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "soak ok" in out, out[-2000:]


def test_shard_invariance_and_determinism():
    """Seeds derive from the GLOBAL env index: 4096 envs in one handle == 4 handles of 1024 envs with
    env_index_base = 0, 1024, ...  (the partitioning of configs[3], on one GPU); and the same seed twice
    gives identical bytes."""
    E, N, steps = 4096, 5, 40

    def run(parts):
        outs = []
        for k in range(parts):
            eng = VecEngine(K.GAME_HARVEST, None, num_envs=E // parts, num_agents=N, seed=123,
                            env_index_base=k * (E // parts))
            obs = eng.reset()
            acc = np.zeros((E // parts, N), np.int64)
            for _ in range(steps):
                obs, rew, _ = eng.step_random()
                acc += rew.cpu().numpy()
            st = eng.get_state()
            outs.append((obs.cpu().numpy(), acc, st["world"], st["pos"], st["orient"]))
        return [np.concatenate([o[i] for o in outs], axis=0) for i in range(5)]

    whole, again, sharded = run(1), run(1), run(4)
    for x, y, z in zip(whole, again, sharded):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)
    assert whole[1].sum() != 0                                # something happened (apples eaten / beams fired)


@pytest.mark.parametrize("game", [K.GAME_HARVEST, K.GAME_CLEANUP])
def test_invariants_over_a_long_rollout(game):
    """1000 steps x 2048 envs: walls never change, no agent ever stands on a wall or leaves the map, the cell
    alphabet stays closed, rewards stay in the range the rules allow, and a reset restores the initial apple /
    waste counts."""
    amap = K.HARVEST_MAP if game == K.GAME_HARVEST else K.CLEANUP_MAP
    base = np.array([[ord(ch) for ch in row] for row in amap], dtype=np.int8)
    wall = base == ord('@')
    E, N = 2048, 5
    eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=9)
    eng.reset()
    st0 = eng.get_state()
    alphabet = set(b" @A") if game == K.GAME_HARVEST else set(b" @AHRS")
    rew_sum = np.zeros((E, N), np.int64)
    for s in range(1000):
        obs, rew, done = eng.step_random()
        if s % 100 == 99:
            r = rew.cpu().numpy()
            assert r.max() <= 1 and r.min() >= -1 - 50 * (N - 1) * 3
            st = eng.get_state()
            assert (st["world"][:, wall] == ord('@')).all() and (st["world"][:, ~wall] != ord('@')).all()
            assert set(np.unique(st["world"]).tolist()) <= alphabet
            pr, pc = st["pos"][..., 0].astype(int), st["pos"][..., 1].astype(int)
            assert (pr > 0).all() and (pr < base.shape[0] - 1).all() and (pc > 0).all() and (pc < base.shape[1] - 1).all()
            assert not wall[pr, pc].any()
            assert (st["t"] == s + 1).all()
            rew_sum += r
    assert rew_sum.max() > 0                                   # apples were eaten somewhere
    eng.reset()
    st1 = eng.get_state()
    np.testing.assert_array_equal((st1["world"] == ord('A')).sum(axis=(1, 2)), (st0["world"] == ord('A')).sum(axis=(1, 2)))
    np.testing.assert_array_equal((st1["world"] == ord('H')).sum(axis=(1, 2)), (st0["world"] == ord('H')).sum(axis=(1, 2)))
    assert eng.status() == 0


def test_step_observation_equals_observe_of_the_stored_state():
    """With keep_beams the state stored in HBM is everything an observation depends on: ssd_observe() after a
    step reproduces that step's observations byte for byte (4096 envs)."""
    eng = VecEngine(K.GAME_CLEANUP, None, num_envs=4096, num_agents=5, seed=2, keep_beams=True)
    eng.reset()
    for _ in range(25):
        obs, _, _ = eng.step_random()
    again = eng.observe(rotate=True)
    assert (obs == again).all().item()
    unrot = eng.observe(rotate=False)
    assert not (unrot == again).all().item()


def _rollout_vs_oracle(game, amap, E, N, seed, steps, step0, every, ring, chains, actions=False, env_base=0, **kw):
    """ssd_rollout_random (actions=True: ssd_rollout_actions with a random action tensor) on a fresh engine against the oracle
    stepped call by call: the last `ring` steps' observations and rewards, and the final state."""
    import torch
    eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=seed, env_index_base=env_base)
    ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=seed, env_base=env_base)
    eng.set_rollout_chains(chains)
    obs = torch.zeros((ring, E, N, 15, 15, 3), dtype=torch.uint8, device="cuda")
    rew = torch.zeros((ring, E, N), dtype=torch.int32, device="cuda")
    done = torch.ones((ring, E, N), dtype=torch.uint8, device="cuda")
    eng.reset()
    ora.reset()
    if actions:
        a_host = np.random.RandomState(seed).randint(-1, 8 if game == K.GAME_HARVEST else 9, size=(steps, E, N)).astype(np.int32)
        a_dev = torch.from_numpy(a_host).cuda()
        eng.rollout_actions(a_dev, steps, obs, rew, done, reset_every=every, step0=step0, **kw)
    else:
        eng.rollout_random(steps, obs, rew, done, reset_every=every, step0=step0, **kw)
    want = {}
    for k in range(step0, step0 + steps):
        if every and k % every == 0:
            ora.reset()
        if actions:
            o_obs, o_rew, _ = ora.step(a_host[k % steps])
        else:
            _, o_obs, o_rew, _ = ora.step_random(want_obs=(k >= step0 + steps - ring))
        want[k] = (o_obs, o_rew)
    g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
    for k in range(step0 + steps - ring, step0 + steps):
        np.testing.assert_array_equal(g_rew[k % ring], want[k][1], err_msg="rewards of step %d" % k)
        assert np.array_equal(g_obs[k % ring], want[k][0]), "observations of step %d differ" % k
    assert not done.any()
    a, b = eng.get_state(), ora.get_state()
    for key in ("world", "pos", "orient", "episode", "t"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    # ... and per-step launches (hipLaunchKernel, the plain kernels) continue from what the rollout left -- whatever dispatch
    # path, kernel variant and workgroup mapping its launches used: a cache must not serve them an older copy of the state
    out = eng.alloc_outputs()
    for s in range(3):
        o, r, _ = eng.step_random(out=out)
        _, o_obs, o_rew, _ = ora.step_random()
        np.testing.assert_array_equal(r.cpu().numpy(), o_rew, err_msg="rewards of per-step launch %d after the rollout" % s)
        assert np.array_equal(o.cpu().numpy(), o_obs), "observations of per-step launch %d after the rollout differ" % s
    # ... and another rollout after those
    eng.rollout_random(ring + 4, obs, rew, done, reset_every=0, step0=step0 + steps)
    for k in range(step0 + steps, step0 + steps + ring + 4):
        _, o_obs, o_rew, _ = ora.step_random()
    k = step0 + steps + ring + 3
    np.testing.assert_array_equal(rew.cpu().numpy()[k % ring], o_rew, err_msg="rewards of the second rollout")
    assert np.array_equal(obs.cpu().numpy()[k % ring], o_obs), "observations of the second rollout differ"
    assert eng.status() == 0
    eng.close()


@pytest.mark.parametrize("game", [K.GAME_HARVEST, K.GAME_CLEANUP])
def test_the_exact_bench_path_at_full_size(game):
    """What bench.py times, at its size: ssd_rollout_random over 4096 envs as 2 chains of 2048 (the library's own dispatch
    queues), ring of one output slot, a full reset every 1000 steps -- 45 steps that cross step 1000 -- against the oracle."""
    amap = K.HARVEST_MAP if game == K.GAME_HARVEST else K.CLEANUP_MAP
    _rollout_vs_oracle(game, amap, 4096, 5, seed=0, steps=45, step0=975, every=1000, ring=1, chains=2)


@pytest.mark.parametrize("cfg", ["harvest_rank7_of_configs3", "cleanup48x36_rank7_of_configs4"])
def test_the_last_ranks_shards_of_the_multi_gpu_configurations(cfg):
    """the env index offsets of the 8-GPU configurations never met the oracle (the PRNG key hashes the GLOBAL
    env index: ssd_config.env_index_base).  Rank 7 of configs[3] (Harvest, 32768 envs over 8 GPUs: envs 28672 .. 32767) and of
    configs[4] (Cleanup 48x36, 10 agents, 16384 envs: envs 14336 .. 16383) as bench.py steps them -- ssd_rollout_random, automatic
    chains, a reset at step 0 -- against the oracle built with the same offset."""
    if cfg.startswith("harvest"):
        _rollout_vs_oracle(K.GAME_HARVEST, K.HARVEST_MAP, 4096, 5, seed=0, steps=10, step0=0, every=1000, ring=1, chains=0, env_base=28672)
    else:
        _rollout_vs_oracle(K.GAME_CLEANUP, K.cleanup_map_48x36(), 2048, 10, seed=0, steps=10, step0=0, every=1000, ring=1, chains=0, env_base=14336)


def test_the_bench_configuration_takes_the_native_dispatch_path():
    """What bench.py reports must be what it meant to measure: at the headline configuration a rollout call goes through the
    library's own dispatch queues with the coherent kernel variant and split rendering (ssd_rollout_path) -- unless the
    environment says otherwise.  A silent fallback to hipLaunchKernel would still pass every parity test."""
    import torch
    eng = VecEngine(K.GAME_HARVEST, None, num_envs=4096, num_agents=5, seed=0)
    out = eng.alloc_outputs()
    ring = tuple(t.unsqueeze(0) for t in out)
    eng.set_rollout_chains(2)
    first = eng.rollout_path()                                       # before the first rollout call: nothing
    assert not any(v for k, v in first.items() if k != "agent_match") and first["agent_match"] == "none", first
    eng.rollout_random(20, *ring, reset_every=1000)
    torch.cuda.synchronize()
    path = eng.rollout_path()
    want_aql = os.environ.get("SSD_AQL", "1") != "0"
    want_coh = want_aql and os.environ.get("SSD_AQL_COHERENT", "1") != "0"
    want_split = want_coh and os.environ.get("SSD_AQL_SPLIT", "1") != "0"
    core = {k: path[k] for k in ("aql", "coherent", "split", "fused", "sync", "chains")}
    assert core == {"aql": want_aql, "coherent": want_coh, "split": want_split, "fused": False, "sync": False, "chains": 2}, path
    assert not path["queue_dropped"] and (path["pool"] >= 2 or not want_aql), path   # (a plain process: the pool's two queues pass their probe)
    # (the HSA agent of the HIP device is found by its PCI address -- the fallbacks, UUID and ordinal, are for boxes that hide it)
    assert path["agent_match"] == ("pci" if want_aql else "none"), path
    # the path must not be named before it is certain -- more argument blocks than the library keeps go through
    # hipLaunchKernel, and ssd_rollout_path says so
    big_rew = torch.empty((1100,) + tuple(ring[1].shape[1:]), dtype=torch.int32, device="cuda")
    eng.rollout_random(2, None, big_rew, None, reset_every=1000, step0=20)   # 2 chains x 1100 slots > 2048 blocks
    assert eng.rollout_path()["aql"] is False and eng.rollout_path()["chains"] == 2, eng.rollout_path()
    eng.rollout_random(20, *ring, reset_every=1000, step0=22)
    assert eng.rollout_path()["aql"] == want_aql
    eng.rollout_random(3, *ring, reset_every=1000, step0=42)              # (short calls are not split)
    assert eng.rollout_path()["split"] is False and eng.rollout_path()["aql"] == want_aql
    eng.rollout_random(8, *ring, reset_every=1000, step0=45, fused=True)
    assert eng.rollout_path()["fused"] and not eng.rollout_path()["aql"]
    torch.cuda.synchronize()
    assert eng.status() == 0


@pytest.mark.parametrize("game_name", ["harvest", "cleanup"])
def test_rollout_auto_lets_the_library_choose_the_form(game_name):
    """SSD_ROLLOUT_AUTO (task 3c): with uint8 observations, index action order and two steps or more the call runs as
    the fused kernel, anything else as without the flag -- ssd_rollout_path says which -- and the results are the oracle's either
    way (device-drawn and caller-supplied actions; a one-step call; float32 observations; an explicit action order)."""
    import torch
    game, amap = (K.GAME_HARVEST, K.HARVEST_MAP) if game_name == "harvest" else (K.GAME_CLEANUP, K.CLEANUP_MAP)
    E, N, na = 2304, 5, (8 if game_name == "harvest" else 9)
    _rollout_vs_oracle(game, amap, E, N, seed=9, steps=17, step0=2, every=11, ring=3, chains=0, fused="auto")
    _rollout_vs_oracle(game, amap, E, N, seed=10, steps=9, step0=0, every=0, ring=2, chains=0, actions=True, fused="auto")
    eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=11)
    ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=11)
    obs = torch.zeros((1, E, N, 15, 15, 3), dtype=torch.uint8, device="cuda")
    rew = torch.zeros((1, E, N), dtype=torch.int32, device="cuda")
    eng.reset()
    ora.reset()
    eng.rollout_random(6, obs, rew, None, fused="auto")
    assert eng.rollout_path()["fused"], eng.rollout_path()
    for _ in range(6):
        _, o_obs, o_rew, _ = ora.step_random()
    assert np.array_equal(obs[0].cpu().numpy(), o_obs) and np.array_equal(rew[0].cpu().numpy(), o_rew)
    eng.rollout_random(1, obs, rew, None, step0=6, fused="auto")             # one step: one launch either way, the per-step kernel
    assert not eng.rollout_path()["fused"], eng.rollout_path()
    _, o_obs, o_rew, _ = ora.step_random()
    assert np.array_equal(obs[0].cpu().numpy(), o_obs) and np.array_equal(rew[0].cpu().numpy(), o_rew)
    obs_f = torch.zeros((1, E, N, 15, 15, 3), dtype=torch.float32, device="cuda")
    eng.rollout_random(4, obs_f, rew, None, step0=7, fused="auto")           # float32 observations: the fused kernel writes none
    assert not eng.rollout_path()["fused"], eng.rollout_path()
    for _ in range(4):
        _, o_obs, o_rew, _ = ora.step_random()
    np.testing.assert_array_equal(obs_f[0].cpu().numpy(), ((o_obs.astype(np.float64) - 128.0) / 255.0).astype(np.float32))
    rng = np.random.RandomState(3)                                           # an explicit action order: the general kernels, per step
    a_host = rng.randint(0, na, size=(3, E, N)).astype(np.int32)
    order = np.stack([np.stack([rng.permutation(N) for _ in range(E)]) for _ in range(3)]).astype(np.uint8)
    eng.rollout_actions(torch.from_numpy(a_host).cuda(), 3, obs, rew, None, step0=0, fused="auto", order=torch.from_numpy(order).cuda())
    assert not eng.rollout_path()["fused"], eng.rollout_path()
    for k in range(3):
        o_obs, o_rew, _ = ora.step(a_host[k], order=order[k])
    assert np.array_equal(obs[0].cpu().numpy(), o_obs) and np.array_equal(rew[0].cpu().numpy(), o_rew)
    assert eng.status() == 0
    eng.close()


@pytest.mark.parametrize("hwq,chains", [(None, 2), ("1", 3), ("2", 2), ("3", 1)])
def test_automatic_chains_stay_within_the_queue_budget(hwq, chains):
    """A process has about four hardware queues before the device time-slices them; the HIP runtime takes up to GPU_MAX_HW_QUEUES of
    them.  The library's pool of dispatch queues is what is left when the process sets that variable (2 otherwise: the rule in
    include/ssd.h), and an automatic chain count never exceeds the pool: 8192 envs are stepped as 2 chains by default, as 3 next to a
    runtime held to 1 queue, as 2 next to one held to 2 (what bench.py does as a rank of a process group), as 1 next to one held to
    3 -- always through the library's own queues, and with the oracle's results.  (A process of its own each: both libraries read the variable once.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import golden_util as G\n"
        "from oracle import pyoracle\n"
        "from sequential_social_dilemma_games_amd import constants as K\n"
        "from sequential_social_dilemma_games_amd.engine import VecEngine\n"
        "E = 8192\n"
        "eng = VecEngine(K.GAME_HARVEST, None, num_envs=E, num_agents=5, seed=11)\n"
        "ora = pyoracle.Oracle(K.GAME_HARVEST, K.HARVEST_MAP, E, 5, G.default_lut(), seed=11)\n"
        "out = eng.alloc_outputs(); ring = tuple(t.unsqueeze(0) for t in out)\n"
        "eng.rollout_random(12, *ring, reset_every=1000, step0=0); torch.cuda.synchronize()\n"
        "ora.reset()\n"
        "for k in range(12): _, o_obs, o_rew, _ = ora.step_random(want_obs=(k == 11))\n"
        "assert np.array_equal(ring[1][0].cpu().numpy(), o_rew) and np.array_equal(ring[0][0].cpu().numpy(), o_obs)\n"
        "p = eng.rollout_path(); assert eng.status() == 0\n"
        "print('PATH', p['aql'], p['chains'])\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ)
    for k in ("GPU_MAX_HW_QUEUES", "SSD_AQL_QUEUES", "SSD_ROLLOUT_CHAINS", "SSD_AQL"):
        env.pop(k, None)
    if hwq:
        env["GPU_MAX_HW_QUEUES"] = hwq
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and ("PATH True %d" % chains) in out, out[-1500:]


@pytest.mark.parametrize("mode", ["calls", "chains", "chains3", "fused", "actions", "actions_fused"])
@pytest.mark.parametrize("cfg", ["harvest25x38", "cleanup48x36"])
def test_enlarged_maps_at_their_bench_sizes(cfg, mode):
    """BASELINE.json's enlarged configurations at the sizes bench.py reports them: Harvest 25x38, 5 agents, 4096 envs (the
    label of configs[1]) and Cleanup 48x36, 10 agents, 2048 envs (configs[4]'s per-GPU share: the LDS-tile stress -- the 64 KB
    clamp of envs_per_block(), the 8 list registers, the pipelining capacity rule), 32 steps with a reset inside, stepped call
    by call, as rollout chains (device-drawn and caller-supplied actions), and as the fused kernel."""
    import torch
    if cfg == "harvest25x38":
        game, amap, E, N = K.GAME_HARVEST, K.harvest_map_25x38(), 4096, 5
    else:
        game, amap, E, N = K.GAME_CLEANUP, K.cleanup_map_48x36(), 2048, 10
    if mode == "calls":
        eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=4)
        ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=4)
        out = eng.alloc_outputs()
        np.testing.assert_array_equal(eng.reset(obs=out[0]).cpu().numpy(), ora.reset())
        for s in range(32):
            if s == 20:
                np.testing.assert_array_equal(eng.reset(obs=out[0]).cpu().numpy(), ora.reset())
            obs, rew, done = eng.step_random(out=out)
            _, o_obs, o_rew, _ = ora.step_random(want_obs=(s % 8 == 7 or s == 31))
            np.testing.assert_array_equal(rew.cpu().numpy(), o_rew, err_msg="rewards, step %d" % s)
            if s % 8 == 7 or s == 31:
                assert np.array_equal(obs.cpu().numpy(), o_obs), "observations differ at step %d" % s
        a, b = eng.get_state(), ora.get_state()
        for k in ("world", "pos", "orient", "episode", "t"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        assert eng.status() == 0
        return
    kw = dict(fused=mode.endswith("fused"), actions=mode.startswith("actions"))
    _rollout_vs_oracle(game, amap, E, N, seed=4, steps=32, step0=3, every=13, ring=2, chains={"chains3": 3, "chains": 2, "actions": 2}.get(mode, 1), **kw)


def test_fuzz_slice_against_the_oracle():
    """A deterministic slice of tools/fuzz_parity.py in the suite: random wall-closed maps (4x4 .. 25x30), 1-13 agents, views
    1x1 .. 21x21, beams 1..8, 1-69 envs, with and without kept beams, each stepped call by call with random action subsets and
    orders, as rollout chains (the library's own dispatch queues), as the fused rollout kernel and with SSD_AUTO_RESET -- all
    against the oracle, bit for bit.  Fixed seed; as many configurations as fit in ~25 s (at least 40), every third one first so
    that the slice spreads over the sequence.
    Round 1's only kept fuzz record ended at `cfg 215: game 1 24x8 N=12 v=8 L=1 E=49 keep=1 seed=561232528 auto step 7`:
    12 Cleanup shooters with beams of length 1 are up to 21 slots of one parallel trace, more than the 8 bits of the per-cell
    slot mask that the then unfinished commit 5d01908 introduced (the commit capped the slots at 8; the log predates it by
    three minutes).  The map of that run cannot be redrawn (the map generator changed since), so that configuration -- game,
    shape, agents, view, beam length, envs, kept beams, engine seed -- is replayed on six maps of its shape."""
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_parity
    fuzz_parity.replay_cfg215(variants=6)
    rng = np.random.RandomState(0)
    cfgs = [fuzz_parity.draw(rng, i) for i in range(216)]
    t0, n = time.time(), 0
    for i in list(range(0, 216, 3)) + list(range(1, 216, 3)) + list(range(2, 216, 3)):
        if n >= 40 and time.time() - t0 > 25.0:
            break
        fuzz_parity.run(cfgs[i], quiet=True)
        n += 1
    assert n >= 40
