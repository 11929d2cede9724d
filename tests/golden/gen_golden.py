#!/usr/bin/env python3
"""Generate the golden transition vectors in tests/golden/*.npz from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed, the reference is never copied.  Usage:  python tests/golden/gen_golden.py

How the reference is run
------------------------
* It is imported unmodified from /root/reference.  Three modules it imports but never uses
  on the arithmetic path are absent here (`ray`, `gym`, `cv2`): tiny `sys.modules` stand-ins
  provide `MultiAgentEnv` (an empty base class), `Box/Dict/Discrete` and an empty `cv2`
  (SURVEY.md section 8c).
* Its two global RNGs are routed to the shared counter PRNG (sequential_social_dilemma_games_amd/prng.py)
  by rebinding the names `np` / `random` inside `social_dilemmas.envs.{map_env,harvest,cleanup}`
  to proxy objects (reference files stay byte-identical):
    np.random.shuffle(list)   -> Fisher-Yates with MOVE draws            (map_env.py:422)
    np.random.rand(1)         -> APPLE / WASTE_COIN draw keyed by the caller's (row, col)
                                                                         (harvest.py:101, cleanup.py:139,150)
    np.random.randint(4)      -> SPAWN_ROT draw                          (map_env.py:666)
    random.shuffle(spawn_pts) -> sort by (SPAWN_POINT draw, cell)        (map_env.py:656)
    random.shuffle(waste_pts) -> sort by (WASTE_ORDER draw, cell)        (cleanup.py:145)

What is recorded
----------------
Every `env.step(...)` call becomes one self-contained transition: (pre-state, episode, t,
actions, action order) -> (post-state, beam overlay, rewards, uint8 observation).  Every
`env.reset()` becomes a reset vector.  Scenarios are (i) the reference's own unit tests
(tests/test_envs.py) restated as scripts, with their literal expectations asserted here,
(ii) random-action rollouts on the shipped and synthetic maps, (iii) crowded small maps
with random action subsets and orders (exercises the overlap quirk of map_env.py:480-483).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from sequential_social_dilemma_games_amd import prng  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402


# ----------------------------------------------------------------------------------------
# stand-ins for absent third-party modules (not on the arithmetic path)
# ----------------------------------------------------------------------------------------
def install_shims():
    import matplotlib
    matplotlib.use("Agg")
    ray = types.ModuleType("ray")
    rllib = types.ModuleType("ray.rllib")
    renv = types.ModuleType("ray.rllib.env")

    class MultiAgentEnv(object):
        pass
    renv.MultiAgentEnv = MultiAgentEnv
    ray.rllib = rllib
    rllib.env = renv
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    class Dict(object):
        def __init__(self, d):
            self.spaces = d
    spaces.Box, spaces.Discrete, spaces.Dict = Box, Discrete, Dict
    gym.spaces = spaces
    for name, mod in (("ray", ray), ("ray.rllib", rllib), ("ray.rllib.env", renv), ("gym", gym),
                      ("gym.spaces", spaces), ("cv2", types.ModuleType("cv2"))):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REFERENCE)


# ----------------------------------------------------------------------------------------
# shared-PRNG context + proxies
# ----------------------------------------------------------------------------------------
class Ctx(object):
    def __init__(self):
        self.seed, self.env, self.episode, self.t, self.W = 0, 0, 0, 0, 1
        self.spawn_calls = self.rot_calls = 0
        self.waste_shuffled = False

    def key(self, stream):
        return prng.phase_key(prng.env_key(self.seed, self.env, self.episode), self.t, stream)

    def begin_reset(self, episode):
        self.episode, self.t = episode, 0
        self.spawn_calls = self.rot_calls = 0
        self.waste_shuffled = False

    def begin_step(self):
        self.t += 1
        self.waste_shuffled = False


CTX = Ctx()


class _NpRandomProxy(object):
    def shuffle(self, lst):                      # map_env.py:422
        pk = CTX.key(prng.S_MOVE)
        for i in range(len(lst) - 1, 0, -1):
            j = prng.randint(prng.draw(pk, i), i + 1)
            lst[i], lst[j] = lst[j], lst[i]

    def rand(self, n):                           # harvest.py:101, cleanup.py:139,150
        assert n == 1
        f = sys._getframe(1)
        assert f.f_code.co_name in ("spawn_apples", "spawn_apples_and_waste"), f.f_code.co_name
        cell = int(f.f_locals["row"]) * CTX.W + int(f.f_locals["col"])
        stream = prng.S_WASTE_COIN if CTX.waste_shuffled else prng.S_APPLE
        return np.array([prng.draw(CTX.key(stream), cell) / 4294967296.0])

    def randint(self, n):                        # map_env.py:666
        agent = CTX.rot_calls
        CTX.rot_calls += 1
        return prng.randint(prng.draw(CTX.key(prng.S_SPAWN_ROT), agent), n)


class NpProxy(object):
    random = _NpRandomProxy()

    def __getattr__(self, name):
        return getattr(np, name)


class RandomProxy(object):
    def shuffle(self, lst):
        who = sys._getframe(1).f_code.co_name
        if who == "spawn_point":                 # map_env.py:656
            agent = CTX.spawn_calls
            CTX.spawn_calls += 1
            pk = CTX.key(prng.S_SPAWN_POINT)
            lst.sort(key=lambda p: (prng.draw(pk, (agent << 16) | (p[0] * CTX.W + p[1])), p[0] * CTX.W + p[1]))
        elif who == "spawn_apples_and_waste":    # cleanup.py:145
            pk = CTX.key(prng.S_WASTE_ORDER)
            lst.sort(key=lambda p: (prng.draw(pk, p[0] * CTX.W + p[1]), p[0] * CTX.W + p[1]))
            CTX.waste_shuffled = True
        else:
            raise AssertionError("unexpected random.shuffle caller " + who)


def import_reference():
    install_shims()
    from social_dilemmas.envs import map_env, harvest, cleanup, agent
    map_env.np = harvest.np = cleanup.np = NpProxy()
    map_env.random = cleanup.random = RandomProxy()
    return map_env, harvest, cleanup, agent


# ----------------------------------------------------------------------------------------
# recording
# ----------------------------------------------------------------------------------------
OCODE = K.ORIENTATION_CODE
ONAME = K.ORIENTATION_NAMES


def grid_to_i8(g):
    return np.array([[ord(ch) for ch in row] for row in g], dtype=np.int8)


def snapshot(env):
    ids = list(env.agents.keys())
    assert ids == ["agent-%d" % i for i in range(len(ids))], ids
    pos = np.array([env.agents[a].get_pos() for a in ids], dtype=np.int16).reshape(len(ids), 2)
    orient = np.array([OCODE[env.agents[a].get_orientation()] for a in ids], dtype=np.uint8)
    beam = np.zeros(env.world_map.shape, dtype=np.int8)
    for r, c, ch in env.beam_pos:
        beam[r, c] = ord(ch)
    return grid_to_i8(env.world_map), beam, pos, orient


def obs_to_u8(obs, ids, V):
    out = np.zeros((len(ids), V, V, 3), dtype=np.uint8)
    for i, a in enumerate(ids):
        x = obs[a] * 255.0 + 128.0
        u = np.rint(x).astype(np.int64)
        assert np.array_equal((u - 128.0) / 255.0, obs[a]), "float64 normalisation is not a pure u8 LUT"
        out[i] = u
    return out


class Recorder(object):
    """Groups transitions by (game, map, N, view_len)."""

    def __init__(self):
        self.groups = {}

    def group(self, game, ascii_map, N, view_len, seed, env_index):
        key = (game, tuple(ascii_map), N, view_len, seed, env_index)
        if key not in self.groups:
            self.groups[key] = dict(game=game, map=list(ascii_map), N=N, view_len=view_len, seed=seed,
                                    env=env_index, steps=[], resets=[])
        return self.groups[key]


REC = Recorder()


class Driver(object):
    """Wraps one reference env; drives it under CTX and records what it does."""

    def __init__(self, env, game, ascii_map, seed=0, env_index=0, view_len=7):
        self.env, self.game, self.map = env, game, list(ascii_map)
        self.seed, self.env_index, self.view_len = seed, env_index, view_len
        self.episode = -1
        self.t = 0

    def _arm(self):
        CTX.seed, CTX.env, CTX.W = self.seed, self.env_index, len(self.map[0])
        CTX.episode, CTX.t = max(self.episode, 0), self.t

    def set_view(self):
        for a in self.env.agents.values():
            a.row_size = a.col_size = self.view_len
            a.view_len = self.view_len

    def reset(self, record=True):
        self._arm()
        self.episode += 1
        self.t = 0
        CTX.begin_reset(self.episode)
        ret = self.env.reset()
        self.set_view()
        # the reference renders reset observations before we could change the view size, so
        # re-render them in reset form (no rotation, map_env.py:232-240)
        obs = self._render(rotate=False)
        if self.view_len == 7 and len(self.env.agents):
            ids = list(self.env.agents.keys())
            assert np.array_equal(obs, obs_to_u8(ret, ids, 15)), "re-rendered reset obs differ from reset()'s"
        if record:
            world, beam, pos, orient = snapshot(self.env)
            g = REC.group(self.game, self.map, len(self.env.agents), self.view_len, self.seed, self.env_index)
            g["resets"].append(dict(episode=self.episode, world=world, pos=pos, orient=orient, obs=obs))

    def _render(self, rotate):
        env = self.env
        ids = list(env.agents.keys())
        V = 2 * self.view_len + 1
        m = env.get_map_with_agents()
        out = {}
        for a in ids:
            ag = env.agents[a]
            ag.grid = m
            rgb = env.map_to_colors(ag.get_state(), env.color_map)
            if rotate:
                rgb = env.rotate_view(ag.orientation, rgb)
            out[a] = (rgb - 128.0) / 255.0
        return obs_to_u8(out, ids, V)

    def step(self, actions, record=True):
        env = self.env
        self.set_view()
        ids = list(env.agents.keys())
        N = len(ids)
        pre = snapshot(env)
        self._arm()
        self.t += 1
        CTX.begin_step()
        assert CTX.t == self.t
        obs, rew, dones, info = env.step(actions)
        assert info == {} and not dones["__all__"]
        if record:
            act = np.full(N, -1, dtype=np.int32)
            order = np.full(N, 0xFF, dtype=np.uint8)
            for k, (aid, a) in enumerate(actions.items()):
                i = ids.index(aid)
                act[i] = a
                order[k] = i
            world, beam, pos, orient = snapshot(env)
            V = 2 * self.view_len + 1
            g = REC.group(self.game, self.map, N, self.view_len, self.seed, self.env_index)
            g["steps"].append(dict(episode=max(self.episode, 0), t=self.t, act=act, order=order,
                                   pre_world=pre[0], pre_pos=pre[2], pre_orient=pre[3],
                                   world=world, beam=beam, pos=pos, orient=orient,
                                   rew=np.array([rew[a] for a in ids], dtype=np.int32),
                                   obs=obs_to_u8(obs, ids, V)))
        return obs, rew, dones, info

    # --- helpers mirroring tests/test_envs.py:695-727, 972-1003 ---
    def move_agent(self, agent_id, new_pos):
        env = self.env
        env.agents[agent_id].set_pos(new_pos)
        env.agents[agent_id].grid = env.get_map_with_agents()
        env.agents[agent_id].update_agent_pos(new_pos)

    def rotate_agent(self, agent_id, new_rot):
        self.env.agents[agent_id].update_agent_rot(new_rot)

    def add_agent(self, agent_cls, agent_id, start_pos, start_orientation):
        env = self.env
        env.agents[agent_id] = agent_cls(agent_id, start_pos, start_orientation, env.get_map_with_agents(),
                                         self.view_len)
        m = env.get_map_with_agents()
        for a in env.agents.values():
            a.grid = m


def random_actions(drv, num_actions):
    N = len(drv.env.agents)
    a = prng.random_actions(drv.seed, [drv.env_index], [max(drv.episode, 0)], drv.t + 1, N, num_actions)[0]
    return {"agent-%d" % i: int(a[i]) for i in range(N)}


def test_map_of(env):
    return np.array(env.test_map)


def expect(env, rows):
    got = test_map_of(env)
    want = np.array([list(r) for r in rows])
    assert np.array_equal(got, want), "\n%s\n!=\n%s" % (got, want)


# ----------------------------------------------------------------------------------------
# scenarios
# ----------------------------------------------------------------------------------------
BASE_MAP_1 = ['@@@@@@@', '@     @', '@     @', '@     @', '@     @', '@     @', '@@@@@@@']
BASE_MAP_2 = ['@@@@@@', '@ P  @', '@    @', '@    @', '@   P@', '@@@@@@']
MINI_HARVEST_MAP = ['@@@@@@', '@ P  @', '@  AA@', '@  AA@', '@  AP@', '@@@@@@']
MINI_CLEANUP_MAP = ['@@@@@@', '@ P  @', '@H BB@', '@R BB@', '@S BP@', '@@@@@@']
FIRING_CLEANUP_MAP = ['@@@@@@', '@    @', '@HHP @', '@RH  @', '@H P @', '@@@@@@']
APPLE_SPAWN_MAP_CLEANUP = ['@@@@@@', '@ P  @', '@  BB@', '@  BB@', '@  BP@', '@@@@@@']
CLEANUP_PROB_MAP = ['@@@@@@', '@    @', '@HHPB@', '@RH B@', '@H PB@', '@@@@@@']
CROWD_MAP_A = ['@@@@@@@', '@PPPPP@', '@PPPPP@', '@PPPPP@', '@@@@@@@']
CROWD_MAP_B = ['@@@@@@@', '@PAPAP@', '@APAPA@', '@PAPAP@', '@@@@@@@']
CROWD_MAP_C = ['@@@@@@@', '@PHBPP@', '@RPBPH@', '@PHBPP@', '@@@@@@@']

A = {v: k for k, v in K.CLEANUP_ACTIONS.items()}


def scen_map_env(ref):
    """tests/test_envs.py TestMapEnv (:157-693) under Harvest rules on apple-free maps."""
    _, harvest, _, agent = ref
    # test_view (:185-317), view_len 2
    d = Driver(harvest.HarvestEnv(BASE_MAP_1, num_agents=0), 0, BASE_MAP_1, seed=11, view_len=2)
    d.reset()
    d.add_agent(agent.HarvestAgent, 'agent-0', [3, 3], 'UP')
    for p in ([3, 3], [2, 3], [1, 3], [3, 2], [3, 1], [4, 3], [5, 3], [3, 4], [3, 5], [5, 5], [1, 1], [5, 1], [1, 5]):
        d.move_agent('agent-0', p)
        d.step({})
        d.step({'agent-0': A['STAY']})
    # test_agent_actions (:319-422): every action x orientation, wall blocking, rotations
    for o in ONAME:
        for p in ([2, 2], [1, 1], [5, 5], [1, 5], [5, 1], [3, 3]):
            for a in range(7):
                d.rotate_agent('agent-0', o)
                d.move_agent('agent-0', p)
                d.step({'agent-0': a})
    d.rotate_agent('agent-0', 'LEFT')
    d.move_agent('agent-0', [2, 2])
    d.step({'agent-0': A['MOVE_LEFT']})
    assert d.env.agents['agent-0'].get_pos().tolist() == [2, 3]
    d.step({'agent-0': A['MOVE_RIGHT']})
    d.step({'agent-0': A['MOVE_UP']})
    assert d.env.agents['agent-0'].get_pos().tolist() == [1, 2]

    # test_agent_conflict (:424-693)
    d = Driver(harvest.HarvestEnv(BASE_MAP_2, num_agents=2), 0, BASE_MAP_2, seed=12, view_len=2)
    d.reset()
    expect(d.env, BASE_MAP_2)
    d.move_agent('agent-0', [3, 3]); d.move_agent('agent-1', [3, 4])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    d.step({'agent-0': A['MOVE_DOWN']})
    d.step({'agent-1': A['MOVE_UP']})
    d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_UP']})
    expect(d.env, ['@@@@@@', '@    @', '@    @', '@  PP@', '@    @', '@@@@@@'])
    d.step({'agent-0': A['MOVE_DOWN']})
    for _ in range(20):
        d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_LEFT']})
        expect(d.env, ['@@@@@@', '@    @', '@   P@', '@   P@', '@    @', '@@@@@@'])
        d.step({'agent-0': A['MOVE_UP'], 'agent-1': A['MOVE_RIGHT']})
    wins = 0
    for _ in range(100):                        # :479-506 random tie-break
        d.move_agent('agent-0', [3, 2]); d.move_agent('agent-1', [3, 4])
        d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_UP']})
        wins += d.env.agents['agent-0'].get_pos().tolist() == [3, 3]
    assert 35 <= wins <= 65, wins
    d.add_agent(agent.HarvestAgent, 'agent-2', [2, 3], 'UP')
    wins = 0
    for _ in range(100):                        # :512-548 three-way
        d.move_agent('agent-0', [3, 2]); d.move_agent('agent-1', [3, 4]); d.move_agent('agent-2', [2, 3])
        d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_UP'], 'agent-2': A['MOVE_RIGHT']})
        wins += d.env.agents['agent-2'].get_pos().tolist() == [3, 3]
    assert 20 <= wins <= 46, wins
    ok = 0
    for _ in range(100):                        # :554-581
        d.move_agent('agent-1', [3, 4]); d.move_agent('agent-2', [2, 2]); d.move_agent('agent-0', [3, 2])
        d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_UP'], 'agent-2': A['MOVE_RIGHT']})
        ok += d.env.agents['agent-2'].get_pos().tolist() != [2, 2]
    assert 35 <= ok <= 65, ok
    d.add_agent(agent.HarvestAgent, 'agent-3', [1, 4], 'UP')
    for _ in range(100):                        # :589-607 two simultaneous conflicts, non-index action order
        d.move_agent('agent-1', [3, 4]); d.move_agent('agent-2', [1, 2])
        d.move_agent('agent-0', [3, 2]); d.move_agent('agent-3', [1, 4])
        d.step({'agent-0': A['MOVE_LEFT'], 'agent-2': A['MOVE_RIGHT'],
                'agent-1': A['MOVE_LEFT'], 'agent-3': A['MOVE_RIGHT']})
    d.move_agent('agent-0', [3, 2]); d.move_agent('agent-2', [2, 2])   # :612-626 nobody can move
    d.move_agent('agent-1', [2, 3]); d.move_agent('agent-3', [3, 3])
    d.step({'agent-0': A['MOVE_LEFT'], 'agent-1': A['MOVE_RIGHT'], 'agent-2': A['MOVE_RIGHT'], 'agent-3': A['MOVE_UP']})
    expect(d.env, ['@@@@@@', '@    @', '@ PP @', '@ PP @', '@    @', '@@@@@@'])
    for _ in range(100):                        # :631-665
        d.move_agent('agent-0', [3, 2]); d.move_agent('agent-2', [2, 2])
        d.move_agent('agent-1', [4, 4]); d.move_agent('agent-3', [3, 3])
        d.step({'agent-0': A['MOVE_RIGHT'], 'agent-2': A['MOVE_RIGHT'], 'agent-3': A['MOVE_UP']})
    d.move_agent('agent-0', [3, 2]); d.move_agent('agent-2', [2, 2])   # :669-681 4-cycle rotates
    d.move_agent('agent-1', [2, 3]); d.move_agent('agent-3', [3, 3])
    d.step({'agent-0': A['MOVE_LEFT'], 'agent-1': A['MOVE_RIGHT'], 'agent-2': A['MOVE_DOWN'], 'agent-3': A['MOVE_UP']})
    assert [d.env.agents['agent-%d' % i].get_pos().tolist() for i in range(4)] == [[2, 2], [3, 3], [2, 3], [3, 2]]
    d.move_agent('agent-0', [2, 1]); d.move_agent('agent-1', [1, 1])   # :685-693 wall + conflict
    d.move_agent('agent-2', [4, 4]); d.move_agent('agent-3', [3, 3])
    before = test_map_of(d.env).copy()
    d.step({'agent-0': A['MOVE_UP'], 'agent-1': A['MOVE_RIGHT']})
    assert np.array_equal(before, test_map_of(d.env))


def scen_harvest(ref):
    """tests/test_envs.py TestHarvestEnv (:730-967)."""
    _, harvest, _, agent = ref
    d = Driver(harvest.HarvestEnv(MINI_HARVEST_MAP, num_agents=0), 0, MINI_HARVEST_MAP, seed=21, view_len=2)
    d.reset()                                   # test_reset :745-755
    expect(d.env, ['@@@@@@', '@    @', '@  AA@', '@  AA@', '@  A @', '@@@@@@'])
    d.env.world_map = np.array([list(r) for r in ['@@@@@@', '@    @', '@    @', '@    @', '@  A @', '@@@@@@']])
    for _ in range(300):                        # test_apple_spawn :757-769
        d.step({})
    assert d.env.count_apples(d.env.test_map) == 5
    d = Driver(harvest.HarvestEnv(MINI_HARVEST_MAP, num_agents=2), 0, MINI_HARVEST_MAP, seed=22, view_len=2)
    d.reset()                                   # :773-802 beams do not block spawning, agents do
    d.move_agent('agent-0', [3, 1]); d.move_agent('agent-1', [3, 3])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    d.step({'agent-1': A['FIRE']})
    d.env.update_map([[2, 1, 'A']])
    d.step({})
    d.step({'agent-1': A['FIRE']})
    d.env.update_map([[3, 1, 'A']])
    d.step({})
    # test_agent_actions :804-850 beam shape
    d = Driver(harvest.HarvestEnv(BASE_MAP_1, num_agents=0), 0, BASE_MAP_1, seed=23, view_len=2)
    d.reset()
    d.add_agent(agent.HarvestAgent, 'agent-0', [2, 2], 'LEFT')
    d.rotate_agent('agent-0', 'UP'); d.move_agent('agent-0', [3, 2])
    d.step({'agent-0': A['FIRE']})
    view = d.env.agents['agent-0'].get_state()
    want = np.array([list('@    '), list('@FF  '), list('@F1  '), list('@FF  '), list('@    ')])
    assert np.array_equal(view, want), view
    d.step({})
    d.rotate_agent('agent-0', 'DOWN'); d.move_agent('agent-0', [3, 2])
    d.step({'agent-0': A['FIRE']})
    view = d.env.agents['agent-0'].get_state()
    want = np.array([list('@    '), list('@ FFF'), list('@ 1FF'), list('@ FFF'), list('@    ')])
    assert np.array_equal(view, want), view
    for o in ONAME:                             # beams from every cell in every direction (walls clip them)
        for r in range(1, 6):
            for c in range(1, 6):
                d.rotate_agent('agent-0', o); d.move_agent('agent-0', [r, c])
                d.step({'agent-0': A['FIRE']})
    d = Driver(harvest.HarvestEnv(MINI_HARVEST_MAP, num_agents=0), 0, MINI_HARVEST_MAP, seed=24, view_len=2)
    d.reset()                                   # :838-850 walking over apples eats them
    d.add_agent(agent.HarvestAgent, 'agent-0', [3, 2], 'RIGHT')
    d.step({'agent-0': A['MOVE_RIGHT']})
    d.step({'agent-0': A['MOVE_LEFT']})
    # test_agent_rewards :852-868
    d = Driver(harvest.HarvestEnv(MINI_HARVEST_MAP, num_agents=2), 0, MINI_HARVEST_MAP, seed=25, view_len=2)
    d.reset()
    d.move_agent('agent-0', [2, 2]); d.move_agent('agent-1', [3, 2])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    _, rew, _, _ = d.step({'agent-0': A['MOVE_DOWN'], 'agent-1': A['MOVE_DOWN']})
    assert rew['agent-0'] == 1 and rew['agent-1'] == 1
    d.rotate_agent('agent-1', 'LEFT')
    _, rew, _, _ = d.step({'agent-1': A['FIRE']})
    assert rew['agent-0'] == -50 and rew['agent-1'] == -1
    # test_agent_conflict :870-915, test_beam_conflict :917-944
    d = Driver(harvest.HarvestEnv(BASE_MAP_2, num_agents=2), 0, BASE_MAP_2, seed=26, view_len=2)
    d.reset()
    d.move_agent('agent-0', [3, 3]); d.move_agent('agent-1', [3, 4])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    d.step({'agent-0': A['MOVE_UP']})
    d.step({'agent-1': A['FIRE']})
    expect(d.env, ['@@@@@@', '@    @', '@FFFF@', '@ FFP@', '@FFFF@', '@@@@@@'])
    d.step({})
    expect(d.env, ['@@@@@@', '@    @', '@    @', '@ P P@', '@    @', '@@@@@@'])
    d.rotate_agent('agent-0', 'DOWN')
    d.step({'agent-0': A['FIRE'], 'agent-1': A['FIRE']})
    d.step({'agent-1': A['FIRE'], 'agent-0': A['FIRE']})
    d.step({})
    d = Driver(harvest.HarvestEnv(MINI_HARVEST_MAP, num_agents=2), 0, MINI_HARVEST_MAP, seed=27, view_len=2)
    d.reset()
    d.move_agent('agent-0', [4, 2]); d.move_agent('agent-1', [4, 4])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    d.step({'agent-1': A['FIRE']})
    expect(d.env, ['@@@@@@', '@    @', '@  AA@', '@FFFF@', '@ FFP@', '@@@@@@'])
    d.step({})
    expect(d.env, ['@@@@@@', '@    @', '@  AA@', '@  AA@', '@ PAP@', '@@@@@@'])


def scen_cleanup(ref):
    """tests/test_envs.py TestCleanupEnv (:1006-1188)."""
    _, _, cleanup, agent = ref
    env = cleanup.CleanupEnv(num_agents=0)
    assert env.potential_waste_area == 119      # test_parameters :1007-1009
    d = Driver(cleanup.CleanupEnv(MINI_CLEANUP_MAP, num_agents=0), 1, MINI_CLEANUP_MAP, seed=31, view_len=2)
    d.reset()                                   # test_reset :1011-1021
    expect(d.env, ['@@@@@@', '@    @', '@H   @', '@R   @', '@S   @', '@@@@@@'])
    # test_cleanup_beam :1023-1089
    d = Driver(cleanup.CleanupEnv(FIRING_CLEANUP_MAP, num_agents=2), 1, FIRING_CLEANUP_MAP, seed=32, view_len=2)
    d.reset()
    d.move_agent('agent-0', [3, 3]); d.move_agent('agent-1', [4, 2])
    d.rotate_agent('agent-0', 'UP')
    d.step({'agent-0': A['CLEAN']})
    expect(d.env, ['@@@@@@', '@    @', '@HCC @', '@RCP @', '@HCC @', '@@@@@@'])
    d.step({})
    d.reset()
    d.move_agent('agent-0', [3, 3]); d.move_agent('agent-1', [4, 2])
    d.env.update_map([[3, 4, 'A']])
    d.rotate_agent('agent-0', 'DOWN')
    d.step({'agent-0': A['CLEAN']})
    d.step({})
    assert d.env.world_map[3, 4] == 'A'
    d.move_agent('agent-1', [2, 2]); d.move_agent('agent-0', [1, 3])
    d.rotate_agent('agent-0', 'RIGHT')
    d.step({'agent-0': A['CLEAN']})
    assert d.env.world_map[2, 2] in 'RH'        # cleaned under the agent (waste may respawn elsewhere)
    d.move_agent('agent-1', [2, 3]); d.move_agent('agent-0', [4, 3])
    d.env.update_map([[2, 2, 'H']]); d.env.update_map([[3, 1, 'H']])
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    d.step({'agent-0': A['CLEAN'], 'agent-1': A['CLEAN']})
    for o in ONAME:                             # CLEAN and FIRE from every free cell, both agents acting
        for r in range(1, 5):
            for c in range(1, 5):
                if [r, c] == [4, 1]:
                    continue
                d.move_agent('agent-1', [4, 1])
                d.rotate_agent('agent-0', o); d.move_agent('agent-0', [r, c])
                d.step({'agent-0': A['CLEAN'], 'agent-1': A['FIRE']})
                d.step({'agent-1': A['CLEAN'], 'agent-0': A['FIRE']})
    # test_firing_beam :1091-1133
    d = Driver(cleanup.CleanupEnv(FIRING_CLEANUP_MAP, num_agents=2), 1, FIRING_CLEANUP_MAP, seed=33, view_len=2)
    d.reset()
    d.move_agent('agent-0', [3, 3]); d.move_agent('agent-1', [4, 2])
    d.rotate_agent('agent-0', 'UP')
    d.step({'agent-0': A['FIRE']})
    expect(d.env, ['@@@@@@', '@    @', '@FFF @', '@FFP @', '@HFF @', '@@@@@@'])
    d.step({})
    expect(d.env, ['@@@@@@', '@    @', '@HH  @', '@RHP @', '@HP  @', '@@@@@@'])
    # test_apple_spawn :1135-1147
    d = Driver(cleanup.CleanupEnv(APPLE_SPAWN_MAP_CLEANUP, num_agents=2), 1, APPLE_SPAWN_MAP_CLEANUP, seed=34, view_len=2)
    d.reset()
    for _ in range(500):
        d.step({})
    tm = test_map_of(d.env)
    assert (tm == 'A').sum() >= 3, tm
    # test_spawn_probabilities :1149-1188
    d = Driver(cleanup.CleanupEnv(CLEANUP_PROB_MAP, num_agents=2), 1, CLEANUP_PROB_MAP, seed=35, view_len=2)
    d.reset()
    assert d.env.compute_permitted_area() == 1 and d.env.potential_waste_area == 5
    assert d.env.current_apple_spawn_prob == 0 and d.env.current_waste_spawn_prob == 0
    d.rotate_agent('agent-0', 'UP'); d.rotate_agent('agent-1', 'UP')
    for _ in range(60):
        d.step({'agent-0': A['CLEAN'], 'agent-1': A['CLEAN']})
    for _ in range(60):
        d.step({'agent-0': int(prng.randint(prng.draw(77, d.t), 9)), 'agent-1': int(prng.randint(prng.draw(78, d.t), 9))})


def scen_rollouts(ref):
    _, harvest, cleanup, _ = ref
    plans = [
        (0, K.HARVEST_MAP, 5, 7, 101, 0, 300),
        (0, K.HARVEST_MAP, 5, 7, 101, 4095, 120),
        (0, K.HARVEST_MAP, 9, 7, 102, 3, 80),
        (1, K.CLEANUP_MAP, 5, 7, 103, 0, 400),
        (1, K.CLEANUP_MAP, 10, 7, 104, 7, 120),
        (0, K.harvest_map_25x38(), 5, 7, 105, 1, 60),
        (1, K.cleanup_map_48x36(), 10, 7, 106, 2, 80),
    ]
    for game, amap, n, v, seed, env_index, steps in plans:
        cls = harvest.HarvestEnv if game == 0 else cleanup.CleanupEnv
        CTX.W = len(amap[0])
        d = Driver(cls(amap, num_agents=n), game, amap, seed=seed, env_index=env_index, view_len=v)
        d.reset()
        for s in range(steps):
            if game == 1 and s % 3 != 2:
                # bias towards CLEAN early on so that waste drops below the 0.4 density threshold
                # and the apple / waste spawn branches are exercised (cleanup.py:156-171)
                a = random_actions(d, 9)
                if d.t < 150:
                    a = {k: (8 if (prng.draw(9, d.t * 16 + i) & 3) else x) for i, (k, x) in enumerate(a.items())}
                d.step(a)
            else:
                d.step(random_actions(d, 8 if game == 0 else 9))
            if s == steps // 2:
                d.reset()


def scen_crowded(ref):
    """Small dense maps, random subsets and orders of actions: chains, swaps, cycles and the
    two-agents-on-one-cell quirk (map_env.py:480-483)."""
    _, harvest, cleanup, _ = ref
    for game, amap, n, seed, steps in ((0, CROWD_MAP_A, 7, 201, 1200), (0, CROWD_MAP_B, 6, 202, 800),
                                       (1, CROWD_MAP_C, 5, 203, 800), (0, CROWD_MAP_A, 12, 204, 400)):
        cls = harvest.HarvestEnv if game == 0 else cleanup.CleanupEnv
        CTX.W = len(amap[0])
        d = Driver(cls(amap, num_agents=n), game, amap, seed=seed, env_index=5, view_len=2)
        d.reset()
        overlaps = 0
        na = 8 if game == 0 else 9
        for s in range(steps):
            u = [prng.draw(seed * 7919 + 13, s * 64 + i) for i in range(2 * n + 1)]
            ids = list(range(n))
            for i in range(n - 1, 0, -1):       # random action order
                j = prng.randint(u[i], i + 1)
                ids[i], ids[j] = ids[j], ids[i]
            keep = n if s % 3 else max(1, prng.randint(u[2 * n], n + 1))
            acts = {}
            for i in ids[:keep]:
                r = prng.randint(u[n + i], 100)
                acts['agent-%d' % i] = int(r % 5) if r < 80 else int(5 + (r - 80) % (na - 5))
            d.step(acts)
            p = [tuple(x) for x in d.env.agent_pos]
            overlaps += len(set(p)) != len(p)
        print("  crowded %s N=%d: %d steps, %d with overlapping agents" % (amap[1], n, steps, overlaps))


# ----------------------------------------------------------------------------------------
def write_fixtures():
    os.makedirs(HERE, exist_ok=True)
    index = []
    for gi, g in enumerate(REC.groups.values()):
        name = "g%02d_%s_%dx%d_n%d_v%d" % (gi, "harvest" if g["game"] == 0 else "cleanup", len(g["map"]),
                                            len(g["map"][0]), g["N"], g["view_len"])
        out = dict(game=np.int32(g["game"]), map=np.array(g["map"]), N=np.int32(g["N"]),
                   view_len=np.int32(g["view_len"]), seed=np.uint64(g["seed"]), env=np.uint32(g["env"]))
        st, rs = g["steps"], g["resets"]
        if st:
            for k in st[0]:
                out["s_" + k] = np.stack([np.asarray(x[k]) for x in st])
        if rs:
            for k in rs[0]:
                out["r_" + k] = np.stack([np.asarray(x[k]) for x in rs])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        index.append((name, len(st), len(rs), os.path.getsize(path)))
    for name, ns, nr, sz in index:
        print("%-40s steps=%5d resets=%3d  %7.1f KB" % (name, ns, nr, sz / 1024.0))
    print("total %.1f KB" % (sum(x[3] for x in index) / 1024.0))


def main():
    ref = import_reference()
    for f in (scen_map_env, scen_harvest, scen_cleanup, scen_rollouts, scen_crowded):
        print(f.__name__)
        f(ref)
    write_fixtures()


if __name__ == "__main__":
    main()
