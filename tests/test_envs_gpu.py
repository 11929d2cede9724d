"""The reference's own unit tests (tests/test_envs.py) restated against the drop-in dict API
(HarvestEnv / CleanupEnv on the HIP engine).  Maps, scenarios and literal expectations are the
reference's; each test cites the lines it restates.  Two mechanical changes: (1) the reference
adds agents by poking `env.agents`, here envs are constructed with the agents a scenario needs
and the agents are moved into place; (2) assertions tied to one Mersenne-Twister stream
(e.g. exactly .53/.47 at :505) become the statistical bounds the neighbouring tests use."""
import numpy as np
import pytest

from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.cleanup import CLEANUP_ACTIONS, CleanupEnv
from sequential_social_dilemma_games_amd.harvest import HARVEST_ACTIONS, HarvestEnv

pytestmark = pytest.mark.gpu

ACTION_MAP = {y: x for x, y in K.BASE_ACTIONS.items()}
HARVEST_ACTION_MAP = {y: x for x, y in HARVEST_ACTIONS.items()}
CLEANUP_ACTION_MAP = {y: x for x, y in CLEANUP_ACTIONS.items()}

BASE_MAP_1 = ['@@@@@@@', '@     @', '@     @', '@     @', '@     @', '@     @', '@@@@@@@']
BASE_MAP_1P = ['@@@@@@@', '@P    @', '@     @', '@     @', '@     @', '@     @', '@@@@@@@']
BASE_MAP_2 = ['@@@@@@', '@ P  @', '@    @', '@    @', '@   P@', '@@@@@@']
BASE_MAP_2P = ['@@@@@@', '@PP  @', '@    @', '@    @', '@  PP@', '@@@@@@']      # four spawn points for 3-4 agents
TEST_MAP_2 = ['@@@@@@', '@    @', '@    @', '@    @', '@  A @', '@@@@@@']
MINI_HARVEST_MAP = ['@@@@@@', '@ P  @', '@  AA@', '@  AA@', '@  AP@', '@@@@@@']
MINI_CLEANUP_MAP = ['@@@@@@', '@ P  @', '@H BB@', '@R BB@', '@S BP@', '@@@@@@']
FIRING_CLEANUP_MAP = ['@@@@@@', '@    @', '@HHP @', '@RH  @', '@H P @', '@@@@@@']
APPLE_SPAWN_MAP_CLEANUP = ['@@@@@@', '@ P  @', '@  BB@', '@  BB@', '@  BP@', '@@@@@@']
CLEANUP_PROB_MAP = ['@@@@@@', '@    @', '@HHPB@', '@RH B@', '@H PB@', '@@@@@@']


def grid(rows):
    return np.array([list(r) for r in rows])


def move_agent(env, agent_id, new_pos):          # tests/test_envs.py:695-702
    env.agents[agent_id].set_pos(new_pos)
    env.agents[agent_id].update_agent_pos(new_pos)


def rotate_agent(env, agent_id, new_rot):        # :704-705
    env.agents[agent_id].update_agent_rot(new_rot)


def pos_of(env, agent_id):
    return env.agents[agent_id].get_pos().tolist()


class TestMapEnv(object):
    def test_step(self):
        """:162-169 every base action runs"""
        env = HarvestEnv(ascii_map=BASE_MAP_2, num_agents=1, seed=1)
        env.reset()
        for i in range(len(ACTION_MAP)):
            obs, rew, dones, info = env.step({'agent-0': i})
        assert obs['agent-0'].shape == (15, 15, 3) and obs['agent-0'].dtype == np.float64
        assert info == {} and dones == {'agent-0': False, '__all__': False}

    def test_walls(self):
        """:171-183"""
        env = HarvestEnv(BASE_MAP_1, num_agents=0, seed=1)
        env.reset()
        for m in (env.base_map, env.world_map):
            for edge in (m[0, :], m[-1, :], m[:, 0], m[:, -1]):
                np.testing.assert_array_equal(edge, np.array(['@'] * 7))

    def test_view(self):
        """:185-317 window crop with '0' padding at every edge and corner (view_len 2)"""
        env = HarvestEnv(BASE_MAP_1P, num_agents=1, seed=1, view_len=2)
        env.reset()
        aid = 'agent-0'
        rotate_agent(env, aid, 'UP')

        def view_at(p):
            move_agent(env, aid, p)
            v = env.agents[aid].get_state().copy()
            v[v == '0'] = ''
            return v

        def rows(*r):
            return np.array([[ch if ch != '0' else '' for ch in row] for row in r])

        np.testing.assert_array_equal(view_at([3, 3]), rows('     ', '     ', '  1  ', '     ', '     '))
        np.testing.assert_array_equal(view_at([2, 3]), rows('@@@@@', '     ', '  1  ', '     ', '     '))
        np.testing.assert_array_equal(view_at([1, 3]), rows('00000', '@@@@@', '  1  ', '     ', '     '))
        np.testing.assert_array_equal(view_at([3, 2]), rows('@    ', '@    ', '@ 1  ', '@    ', '@    '))
        np.testing.assert_array_equal(view_at([3, 1]), rows('0@   ', '0@   ', '0@1  ', '0@   ', '0@   '))
        np.testing.assert_array_equal(view_at([4, 3]), rows('     ', '     ', '  1  ', '     ', '@@@@@'))
        np.testing.assert_array_equal(view_at([5, 3]), rows('     ', '     ', '  1  ', '@@@@@', '00000'))
        np.testing.assert_array_equal(view_at([3, 4]), rows('    @', '    @', '  1 @', '    @', '    @'))
        np.testing.assert_array_equal(view_at([3, 5]), rows('   @0', '   @0', '  1@0', '   @0', '   @0'))
        np.testing.assert_array_equal(view_at([5, 5]), rows('   @0', '   @0', '  1@0', '@@@@0', '00000'))
        # the same windows as rendered by the kernel (colours of the characters above)
        move_agent(env, aid, [5, 5])
        obs, _, _, _ = env.step({})
        want = env.map_to_colors(env.agents[aid].get_state(), env.color_map)
        np.testing.assert_array_equal(obs[aid], (want - 128.0) / 255.0)

    def test_agent_actions(self):
        """:319-422 action x orientation table, walls, rotations"""
        env = HarvestEnv(BASE_MAP_1P, num_agents=1, seed=1, view_len=2)
        env.reset()
        aid = 'agent-0'
        move_agent(env, aid, [2, 2])
        table = {'LEFT': ([2, 3], [2, 2], [1, 2], [2, 2]), 'UP': ([1, 2], [2, 2], [2, 1], [2, 2]),
                 'DOWN': ([3, 2], [2, 2], [2, 3], [2, 2]), 'RIGHT': ([2, 1], [2, 2], [3, 2], [2, 2])}
        for facing in ('LEFT', 'UP', 'DOWN', 'RIGHT'):
            rotate_agent(env, aid, facing)
            for name, want in zip(('MOVE_LEFT', 'MOVE_RIGHT', 'MOVE_UP', 'MOVE_DOWN'), table[facing]):
                env.step({aid: ACTION_MAP[name]})
                assert pos_of(env, aid) == want, (facing, name)
        env.step({aid: ACTION_MAP['STAY']})
        assert pos_of(env, aid) == [2, 2] and env.test_map[2, 2] == 'P'
        # walls (:376-400)
        rotate_agent(env, aid, 'UP')
        move_agent(env, aid, [1, 1])
        env.step({aid: ACTION_MAP['MOVE_UP']}); assert pos_of(env, aid) == [1, 1]
        env.step({aid: ACTION_MAP['MOVE_LEFT']}); assert pos_of(env, aid) == [1, 1]
        move_agent(env, aid, [4, 4])
        for a in ('MOVE_RIGHT', 'MOVE_DOWN', 'MOVE_RIGHT'):
            env.step({aid: ACTION_MAP[a]})
        assert pos_of(env, aid) == [5, 5]
        env.step({aid: ACTION_MAP['MOVE_DOWN']}); assert pos_of(env, aid) == [5, 5]
        env.step({aid: ACTION_MAP['MOVE_LEFT']}); env.step({aid: ACTION_MAP['MOVE_DOWN']})
        assert pos_of(env, aid) == [4, 5]
        move_agent(env, aid, [5, 4]); env.step({aid: ACTION_MAP['MOVE_RIGHT']}); assert pos_of(env, aid) == [5, 4]
        move_agent(env, aid, [1, 2]); env.step({aid: ACTION_MAP['MOVE_LEFT']}); assert pos_of(env, aid) == [1, 2]
        move_agent(env, aid, [2, 1]); env.step({aid: ACTION_MAP['MOVE_UP']}); assert pos_of(env, aid) == [2, 1]
        # rotations (:402-422)
        rotate_agent(env, aid, 'UP')
        for want in ('RIGHT', 'DOWN', 'LEFT', 'UP'):
            env.step({aid: ACTION_MAP['TURN_CLOCKWISE']})
            assert env.agents[aid].get_orientation() == want
        for want in ('LEFT', 'DOWN', 'RIGHT', 'UP'):
            env.step({aid: ACTION_MAP['TURN_COUNTERCLOCKWISE']})
            assert env.agents[aid].get_orientation() == want

    def test_agent_conflict_two_agents(self):
        """:424-506"""
        env = HarvestEnv(ascii_map=BASE_MAP_2, num_agents=2, seed=3, view_len=2)
        env.reset()
        np.testing.assert_array_equal(env.base_map, env.test_map)       # both spawn points taken
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [3, 4])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-0': ACTION_MAP['MOVE_DOWN']})
        env.step({'agent-1': ACTION_MAP['MOVE_UP']})
        assert pos_of(env, 'agent-0') == [3, 3] and pos_of(env, 'agent-1') == [3, 4]
        env.step({'agent-0': ACTION_MAP['MOVE_DOWN'], 'agent-1': ACTION_MAP['MOVE_UP']})     # no walking through each other
        assert pos_of(env, 'agent-0') == [3, 3] and pos_of(env, 'agent-1') == [3, 4]
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@    @', '@  PP@', '@    @', '@@@@@@']))
        env.step({'agent-0': ACTION_MAP['MOVE_DOWN']})
        for _ in range(25):                                              # follow into a vacated cell (:463-474)
            env.step({'agent-0': ACTION_MAP['MOVE_DOWN'], 'agent-1': ACTION_MAP['MOVE_LEFT']})
            np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@   P@', '@   P@', '@    @', '@@@@@@']))
            env.step({'agent-0': ACTION_MAP['MOVE_UP'], 'agent-1': ACTION_MAP['MOVE_RIGHT']})
        wins = 0
        for _ in range(100):                                             # random tie-break (:479-506)
            move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-1', [3, 4])
            env.step({'agent-0': ACTION_MAP['MOVE_DOWN'], 'agent-1': ACTION_MAP['MOVE_UP']})
            wins += pos_of(env, 'agent-0') == [3, 3]
            e1 = grid(['@@@@@@', '@    @', '@    @', '@ PP @', '@    @', '@@@@@@'])
            e2 = grid(['@@@@@@', '@    @', '@    @', '@  PP@', '@    @', '@@@@@@'])
            assert np.array_equal(env.test_map, e1) or np.array_equal(env.test_map, e2)
        assert 30 <= wins <= 70

    def test_agent_conflict_three_and_four_agents(self):
        """:508-693"""
        env = HarvestEnv(ascii_map=BASE_MAP_2P, num_agents=3, seed=4, view_len=2)
        env.reset()
        for a in env.agents:
            rotate_agent(env, a, 'UP')
        wins = 0
        for _ in range(100):                                             # :512-548 three-way, about 1/3 each
            move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-1', [3, 4]); move_agent(env, 'agent-2', [2, 3])
            env.step({'agent-0': ACTION_MAP['MOVE_DOWN'], 'agent-1': ACTION_MAP['MOVE_UP'], 'agent-2': ACTION_MAP['MOVE_RIGHT']})
            wins += pos_of(env, 'agent-2') == [3, 3]
        assert 20 <= wins <= 47
        ok = 0
        for _ in range(100):                                             # :554-581 moving into a contested agent
            move_agent(env, 'agent-1', [3, 4]); move_agent(env, 'agent-2', [2, 2]); move_agent(env, 'agent-0', [3, 2])
            env.step({'agent-0': ACTION_MAP['MOVE_DOWN'], 'agent-1': ACTION_MAP['MOVE_UP'], 'agent-2': ACTION_MAP['MOVE_RIGHT']})
            if pos_of(env, 'agent-2') == [2, 2]:
                np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@ P  @', '@ PP @', '@    @', '@@@@@@']))
            else:
                ok += 1
                np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@    @', '@ PPP@', '@    @', '@@@@@@']))
        assert 35 <= ok <= 65

        env = HarvestEnv(ascii_map=BASE_MAP_2P, num_agents=4, seed=5, view_len=2)
        env.reset()
        for a in env.agents:
            rotate_agent(env, a, 'UP')
        w0 = w1 = 0
        for _ in range(100):                                             # :589-607 two conflicts at once
            move_agent(env, 'agent-1', [3, 4]); move_agent(env, 'agent-2', [1, 2])
            move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-3', [1, 4])
            env.step({'agent-0': ACTION_MAP['MOVE_LEFT'], 'agent-2': ACTION_MAP['MOVE_RIGHT'],
                      'agent-1': ACTION_MAP['MOVE_LEFT'], 'agent-3': ACTION_MAP['MOVE_RIGHT']})
            w0 += pos_of(env, 'agent-0') == [2, 2]
            w1 += pos_of(env, 'agent-1') == [2, 4]
        assert 35 <= w0 <= 65 and 35 <= w1 <= 65
        move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-2', [2, 2])          # :612-626 gridlock
        move_agent(env, 'agent-1', [2, 3]); move_agent(env, 'agent-3', [3, 3])
        env.step({'agent-0': ACTION_MAP['MOVE_LEFT'], 'agent-1': ACTION_MAP['MOVE_RIGHT'],
                  'agent-2': ACTION_MAP['MOVE_RIGHT'], 'agent-3': ACTION_MAP['MOVE_UP']})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@ PP @', '@ PP @', '@    @', '@@@@@@']))
        wins = 0
        for _ in range(100):                                             # :631-665
            move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-2', [2, 2])
            move_agent(env, 'agent-1', [4, 4]); move_agent(env, 'agent-3', [3, 3])
            env.step({'agent-0': ACTION_MAP['MOVE_RIGHT'], 'agent-2': ACTION_MAP['MOVE_RIGHT'], 'agent-3': ACTION_MAP['MOVE_UP']})
            if pos_of(env, 'agent-2') == [3, 2]:
                wins += 1
                np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@    @', '@ PP @', '@ P P@', '@@@@@@']))
            else:
                np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@ P  @', '@ P  @', '@ P P@', '@@@@@@']))
        assert 35 <= wins <= 65
        move_agent(env, 'agent-0', [3, 2]); move_agent(env, 'agent-2', [2, 2])          # :669-681 a 4-cycle rotates
        move_agent(env, 'agent-1', [2, 3]); move_agent(env, 'agent-3', [3, 3])
        env.step({'agent-0': ACTION_MAP['MOVE_LEFT'], 'agent-1': ACTION_MAP['MOVE_RIGHT'],
                  'agent-2': ACTION_MAP['MOVE_DOWN'], 'agent-3': ACTION_MAP['MOVE_UP']})
        assert [pos_of(env, 'agent-%d' % i) for i in range(4)] == [[2, 2], [3, 3], [2, 3], [3, 2]]
        move_agent(env, 'agent-0', [2, 1]); move_agent(env, 'agent-1', [1, 1])          # :685-693 wall + conflict
        move_agent(env, 'agent-2', [4, 4]); move_agent(env, 'agent-3', [3, 3])
        before = env.test_map.copy()
        env.step({'agent-0': ACTION_MAP['MOVE_UP'], 'agent-1': ACTION_MAP['MOVE_RIGHT']})
        np.testing.assert_array_equal(env.test_map, before)


class TestHarvestEnv(object):
    def test_step(self):
        """:736-743 (the reference's version errors on agent.action_space; the env's space is used here)"""
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=1, seed=1)
        env.reset()
        assert env.action_space.n == 8
        for i in range(env.action_space.n):
            env.step({'agent-0': i})
        with pytest.raises(KeyError):
            env.step({'agent-0': 8})

    def test_reset(self):
        """:745-755"""
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=0, seed=1)
        env.reset()
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@  AA@', '@  AA@', '@  A @', '@@@@@@']))

    def test_apple_spawn(self):
        """:757-802"""
        env = HarvestEnv(MINI_HARVEST_MAP, num_agents=0, seed=2)
        env.reset()
        env.world_map = grid(TEST_MAP_2)
        for _ in range(300):
            env.step({})
        assert env.count_apples(env.test_map) == 5
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=2, seed=2, view_len=2)
        env.reset()
        move_agent(env, 'agent-0', [3, 1]); move_agent(env, 'agent-1', [3, 3])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-1': HARVEST_ACTION_MAP['FIRE']})
        env.update_map([[2, 1, 'A']])
        env.step({})
        want = grid(['@@@@@@', '@    @', '@A AA@', '@P PA@', '@  A @', '@@@@@@'])
        np.testing.assert_array_equal(env.test_map, want)
        env.step({'agent-1': HARVEST_ACTION_MAP['FIRE']})
        env.update_map([[3, 1, 'A']])                                    # an apple under an agent stays hidden under it
        env.step({})
        np.testing.assert_array_equal(env.test_map, want)

    def test_agent_actions(self):
        """:804-850 beam shape and apple consumption"""
        env = HarvestEnv(BASE_MAP_1P, num_agents=1, seed=1, view_len=2)
        env.reset()
        aid = 'agent-0'
        rotate_agent(env, aid, 'UP'); move_agent(env, aid, [3, 2])
        env.step({aid: HARVEST_ACTION_MAP['FIRE']})
        np.testing.assert_array_equal(env.agents[aid].get_state(), grid(['@    ', '@FF  ', '@F1  ', '@FF  ', '@    ']))
        env.step({})
        rotate_agent(env, aid, 'DOWN'); move_agent(env, aid, [3, 2])
        env.step({aid: HARVEST_ACTION_MAP['FIRE']})
        np.testing.assert_array_equal(env.agents[aid].get_state(), grid(['@    ', '@ FFF', '@ 1FF', '@ FFF', '@    ']))
        env = HarvestEnv(MINI_HARVEST_MAP, num_agents=1, seed=1, view_len=2)
        env.reset()
        move_agent(env, aid, [3, 2]); rotate_agent(env, aid, 'RIGHT')
        env.step({aid: HARVEST_ACTION_MAP['MOVE_RIGHT']})
        env.step({aid: HARVEST_ACTION_MAP['MOVE_LEFT']})
        view = env.agents[aid].get_state()
        assert view[2, 2] == '1' and view[2, 3] in ' A'                  # the apple at [3, 3] was eaten (it may respawn)
        assert pos_of(env, aid) == [3, 2]

    def test_agent_rewards(self):
        """:852-868 +1 apple, -1 fire, -50 hit"""
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=2, seed=1)
        env.reset()
        move_agent(env, 'agent-0', [2, 2]); move_agent(env, 'agent-1', [3, 2])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        _, rew, _, _ = env.step({'agent-0': HARVEST_ACTION_MAP['MOVE_DOWN'], 'agent-1': HARVEST_ACTION_MAP['MOVE_DOWN']})
        assert rew == {'agent-0': 1, 'agent-1': 1}
        rotate_agent(env, 'agent-1', 'LEFT')
        _, rew, _, _ = env.step({'agent-1': HARVEST_ACTION_MAP['FIRE']})
        assert rew == {'agent-0': -50, 'agent-1': -1}

    def test_agent_conflict(self):
        """:870-915 beams cover agents for exactly one step"""
        env = HarvestEnv(ascii_map=BASE_MAP_2, num_agents=2, seed=1)
        env.reset()
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [3, 4])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-0': HARVEST_ACTION_MAP['MOVE_UP']})
        env.step({'agent-1': HARVEST_ACTION_MAP['FIRE']})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@FFFF@', '@ FFP@', '@FFFF@', '@@@@@@']))
        env.step({})
        clear = grid(['@@@@@@', '@    @', '@    @', '@ P P@', '@    @', '@@@@@@'])
        np.testing.assert_array_equal(env.test_map, clear)
        rotate_agent(env, 'agent-0', 'DOWN')
        env.step({'agent-0': HARVEST_ACTION_MAP['FIRE'], 'agent-1': HARVEST_ACTION_MAP['FIRE']})
        env.step({})
        np.testing.assert_array_equal(env.test_map, clear)

    def test_beam_conflict(self):
        """:917-944"""
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=2, seed=1)
        env.reset()
        move_agent(env, 'agent-0', [4, 2]); move_agent(env, 'agent-1', [4, 4])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-1': HARVEST_ACTION_MAP['FIRE']})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@  AA@', '@FFFF@', '@ FFP@', '@@@@@@']))
        env.step({})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@  AA@', '@  AA@', '@ PAP@', '@@@@@@']))

    def test_rotation(self):
        """:946-967 rotate_view = rot90 k=0/1/2/3 for UP/LEFT/DOWN/RIGHT, on the host helper and in the kernel"""
        env = HarvestEnv(ascii_map=MINI_HARVEST_MAP, num_agents=2, seed=1)
        m = np.array([[[1, 1, 1], [2, 2, 2]], [[3, 3, 3], [4, 4, 4]]])
        np.testing.assert_array_equal(env.rotate_view('LEFT', m), [[[2, 2, 2], [4, 4, 4]], [[1, 1, 1], [3, 3, 3]]])
        np.testing.assert_array_equal(env.rotate_view('UP', m), m)
        np.testing.assert_array_equal(env.rotate_view('DOWN', m), [[[4, 4, 4], [3, 3, 3]], [[2, 2, 2], [1, 1, 1]]])
        np.testing.assert_array_equal(env.rotate_view('RIGHT', m), [[[3, 3, 3], [1, 1, 1]], [[4, 4, 4], [2, 2, 2]]])
        env.reset()
        for facing in ('UP', 'LEFT', 'DOWN', 'RIGHT'):
            rotate_agent(env, 'agent-0', facing)
            obs, _, _, _ = env.step({})
            want = env.rotate_view(facing, env.map_to_colors(env.agents['agent-0'].get_state(), env.color_map))
            np.testing.assert_array_equal(obs['agent-0'], (want - 128.0) / 255.0)


class TestCleanupEnv(object):
    def test_parameters(self):
        """:1007-1009"""
        env = CleanupEnv(num_agents=0, seed=1)
        assert env.potential_waste_area == 119

    def test_reset(self):
        """:1011-1021"""
        env = CleanupEnv(ascii_map=MINI_CLEANUP_MAP, num_agents=0, seed=1)
        env.reset()
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@H   @', '@R   @', '@S   @', '@@@@@@']))

    def test_cleanup_beam(self):
        """:1023-1089"""
        env = CleanupEnv(ascii_map=FIRING_CLEANUP_MAP, num_agents=2, seed=1)
        env.reset()
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [4, 2])
        rotate_agent(env, 'agent-0', 'UP')
        env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN']})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@HCC @', '@RCP @', '@HCC @', '@@@@@@']))
        assert env.world_map[2, 2] == 'R' and env.world_map[3, 2] == 'R'      # cleaned; the beam stopped at the first waste cell
        env.reset()
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [4, 2])
        env.update_map([[3, 4, 'A']])
        rotate_agent(env, 'agent-0', 'DOWN')
        env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN']})
        env.step({})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@HH  @', '@RHPA@', '@HP  @', '@@@@@@']))
        move_agent(env, 'agent-1', [2, 2]); move_agent(env, 'agent-0', [1, 3])           # clean under an agent
        rotate_agent(env, 'agent-0', 'RIGHT')
        env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN']})
        assert (2, 2, 'C') in env.beam_pos                               # the beam covered the agent's cell and cleaned it
        move_agent(env, 'agent-1', [2, 3]); move_agent(env, 'agent-0', [4, 3])           # beams compose within a step
        env.update_map([[2, 2, 'H']]); env.update_map([[3, 1, 'H']])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN'], 'agent-1': CLEANUP_ACTION_MAP['CLEAN']})
        # agent-0 cleared [3, 2] first, so agent-1's beam was no longer blocked there and reached [3, 1]
        # (the reference pins random.seed(7) so that no waste respawns on it in the same step; the beam
        # overlay shows the same thing deterministically)
        assert (3, 1, 'C') in env.beam_pos and env.world_map[3, 2] in 'RH'

    def test_firing_beam(self):
        """:1091-1133 FIRE passes over waste and cleans nothing"""
        env = CleanupEnv(ascii_map=FIRING_CLEANUP_MAP, num_agents=2, seed=1)
        env.reset()
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [4, 2])
        rotate_agent(env, 'agent-0', 'UP')
        env.step({'agent-0': CLEANUP_ACTION_MAP['FIRE']})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@FFF @', '@FFP @', '@HFF @', '@@@@@@']))
        env.step({})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@HH  @', '@RHP @', '@HP  @', '@@@@@@']))
        env.reset()
        move_agent(env, 'agent-0', [3, 3]); move_agent(env, 'agent-1', [4, 2])
        env.update_map([[3, 4, 'A']])
        rotate_agent(env, 'agent-0', 'DOWN')
        env.step({'agent-0': CLEANUP_ACTION_MAP['FIRE']})
        env.step({})
        np.testing.assert_array_equal(env.test_map, grid(['@@@@@@', '@    @', '@HH  @', '@RHPA@', '@HP  @', '@@@@@@']))

    def test_apple_spawn(self):
        """:1135-1147 every 'B' cell not under an agent fills up"""
        env = CleanupEnv(ascii_map=APPLE_SPAWN_MAP_CLEANUP, num_agents=2, seed=1)
        env.reset()
        for _ in range(500):
            env.step({})
        tm = env.test_map
        for r, c in env.apple_points:
            assert tm[r, c] in 'AP'
        assert (tm == 'P').sum() == 2

    def test_spawn_probabilities(self):
        """:1149-1188"""
        env = CleanupEnv(ascii_map=CLEANUP_PROB_MAP, num_agents=2, seed=1)
        env.reset()
        assert env.compute_permitted_area() == 1 and env.potential_waste_area == 5
        assert np.isclose(env.current_apple_spawn_prob, 0) and np.isclose(env.current_waste_spawn_prob, 0)
        move_agent(env, 'agent-0', [2, 3]); move_agent(env, 'agent-1', [4, 3])
        rotate_agent(env, 'agent-0', 'UP'); rotate_agent(env, 'agent-1', 'UP')
        env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN'], 'agent-1': CLEANUP_ACTION_MAP['CLEAN']})
        assert np.isclose(env.current_waste_spawn_prob, 0.5)
        for _ in range(200):
            env.step({'agent-0': CLEANUP_ACTION_MAP['CLEAN'], 'agent-1': CLEANUP_ACTION_MAP['CLEAN']})
            if env.compute_permitted_area() == 4:
                break
        assert env.compute_permitted_area() == 4
        env.compute_probabilities()
        assert np.isclose(env.current_apple_spawn_prob, 0.025)
        # waste can spawn under an agent (:1184-1188): park an agent on a clean waste cell until it turns to waste
        env.update_map([[2, 1, 'R'], [2, 2, 'R'], [3, 2, 'R'], [4, 1, 'R']])
        move_agent(env, 'agent-0', [2, 2]); move_agent(env, 'agent-1', [1, 4])
        for _ in range(400):
            env.update_map([[2, 1, 'R'], [3, 1, 'R'], [3, 2, 'R'], [4, 1, 'R']])
            env.step({})
            if env.world_map[2, 2] == 'H':
                break
        assert env.world_map[2, 2] == 'H' and pos_of(env, 'agent-0') == [2, 2]


def test_return_agent_actions_observation_dict():
    """map_env.py:201-205,242-246: curr_obs / other_agent_actions / visible_agents (always ones, :767)"""
    env = HarvestEnv(num_agents=3, return_agent_actions=True, seed=9)
    obs = env.reset()
    for aid in env.agents:
        assert set(obs[aid]) == {"curr_obs", "other_agent_actions", "visible_agents"}
        np.testing.assert_array_equal(obs[aid]["other_agent_actions"], [0, 0])
        np.testing.assert_array_equal(obs[aid]["visible_agents"], [1, 1])
        assert obs[aid]["other_agent_actions"].dtype == np.int64
    obs, _, _, _ = env.step({'agent-0': 1, 'agent-1': 5, 'agent-2': 7})
    np.testing.assert_array_equal(obs['agent-0']["other_agent_actions"], [5, 7])
    np.testing.assert_array_equal(obs['agent-1']["other_agent_actions"], [1, 7])
    np.testing.assert_array_equal(obs['agent-2']["other_agent_actions"], [1, 5])
    assert obs['agent-0']["curr_obs"].shape == (15, 15, 3)


def test_seeding_through_numpy_global_rng_is_reproducible():
    def run():
        np.random.seed(123)
        env = CleanupEnv(num_agents=5)
        out = [env.reset()['agent-3'].copy()]
        for t in range(10):
            obs, rew, _, _ = env.step({'agent-%d' % i: (3 * t + i) % 9 for i in range(5)})
            out.append(obs['agent-3'].copy())
        return np.stack(out), env.world_map.copy()
    a, wa = run()
    b, wb = run()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(wa, wb)
    assert a.min() >= (0 - 128.0) / 255.0 and a.max() <= (255 - 128.0) / 255.0
