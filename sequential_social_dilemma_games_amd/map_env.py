"""Drop-in mirror of the reference's `MapEnv` (social_dilemmas/envs/map_env.py:60) on the MI355X engine.

Same constructor arguments, same `reset()` / `step({agent_id: int})` dict API, same secondary
surface the reference's callers and tests use (`agents`, `agent_pos`, `world_map`, `base_map`,
`test_map`, `beam_pos`, `update_map`, `map_to_colors`, `get_map_with_agents`, `render`,
`rotate_view`, ...).  All state lives in the HBM of one GPU (a `VecEngine` with one env); every
method that reads or edits state is a short host round trip through the C ABI, and `step()` is one
launch of the fused HIP kernel.  For throughput use `VecEngine` / `vector_env.SSDVectorEnv`, which
step thousands of envs per launch: this class exists so that RLlib-style code and the reference's
tests run unchanged.

Differences from the reference, all forced by moving the state off the Python heap:
  * randomness comes from the engine's counter-based PRNG (prng.py); `seed=None` draws the seed
    from NumPy's global generator, so `np.random.seed(s)` before construction still makes a run
    reproducible;
  * the number of agents is fixed at construction (the reference's tests add agents by poking
    `env.agents`; here construct the env with the agents you need and move them);
  * `world_map` returns a snapshot: edit cells with `update_map()` or assign a whole grid to
    `env.world_map`;
  * the constructor already builds the map (the reference leaves it blank until `reset()`).
"""
import numpy as np

from . import constants as K
from .engine import VecEngine

try:  # RLlib present: be a real MultiAgentEnv (map_env.py:9)
    from ray.rllib.env import MultiAgentEnv
except Exception:  # pragma: no cover - ray is absent in the build image
    class MultiAgentEnv(object):
        pass

try:
    from gym.spaces import Box, Dict, Discrete
except Exception:  # pragma: no cover - gym is absent in the build image
    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    class Dict(object):
        def __init__(self, spaces):
            self.spaces = spaces

# map_env.py:11-22 (FIRE / CLEAN lengths are injected by harvest.py:11 / cleanup.py:11-12)
ACTIONS = {'MOVE_LEFT': [-1, 0], 'MOVE_RIGHT': [1, 0], 'MOVE_UP': [0, -1], 'MOVE_DOWN': [0, 1], 'STAY': [0, 0],
           'TURN_CLOCKWISE': [[0, -1], [1, 0]], 'TURN_COUNTERCLOCKWISE': [[0, 1], [-1, 0]],
           'FIRE': K.BEAM_LEN, 'CLEAN': K.BEAM_LEN}
ORIENTATIONS = {'LEFT': [-1, 0], 'RIGHT': [1, 0], 'UP': [0, -1], 'DOWN': [0, 1]}
DEFAULT_COLOURS = {k: list(v) for k, v in K.DEFAULT_COLOURS.items()}

# (u8 - 128.0) / 255.0 for every byte value: the reference's float64 normalisation (map_env.py:199,240)
_NORMALISE = (np.arange(256, dtype=np.float64) - 128.0) / 255.0


def return_view(grid, pos, row_size, col_size):
    """Window of `grid` centred on pos, '0' outside the map (utility_funcs.py:59-114)."""
    r, c = int(pos[0]), int(pos[1])
    H, W = grid.shape
    view = np.full((2 * row_size + 1, 2 * col_size + 1), '0', dtype=grid.dtype)
    r0, r1, c0, c1 = max(r - row_size, 0), min(r + row_size, H - 1), max(c - col_size, 0), min(c + col_size, W - 1)
    if r0 <= r1 and c0 <= c1:
        view[r0 - (r - row_size):r1 - (r - row_size) + 1, c0 - (c - col_size):c1 - (c - col_size) + 1] = grid[r0:r1 + 1, c0:c1 + 1]
    return view


class Agent(object):
    """View of one agent's state held by the engine (reference: agent.py:16)."""

    action_table = K.BASE_ACTIONS

    def __init__(self, env, index):
        self._env, self._i = env, index
        self.agent_id = 'agent-' + str(index)
        self.row_size = self.col_size = self.view_len = env.view_len

    # --- state ---
    @property
    def pos(self):
        return self._env._state()["pos"][0, self._i].astype(np.int64)

    @property
    def orientation(self):
        return K.ORIENTATION_NAMES[int(self._env._state()["orient"][0, self._i])]

    def get_pos(self):
        return self.pos

    def get_orientation(self):
        return self.orientation

    def set_pos(self, new_pos):
        pos = self._env._state()["pos"].copy()
        pos[0, self._i] = [int(new_pos[0]), int(new_pos[1])]
        self._env._engine.set_state(pos=pos)
        self._env._dirty()

    def update_agent_pos(self, new_pos):
        """agent.py:115-134: walls block (the agent then keeps its position)."""
        old = self.pos
        if self._env.world_map[int(new_pos[0]), int(new_pos[1])] != '@':
            self.set_pos(new_pos)
        return self.pos, old

    def set_orientation(self, new_orientation):
        orient = self._env._state()["orient"].copy()
        orient[0, self._i] = K.ORIENTATION_CODE[new_orientation]
        self._env._engine.set_state(orient=orient)
        self._env._dirty()

    update_agent_rot = set_orientation

    # --- views ---
    def get_map(self):
        return self._env.get_map_with_agents()

    @property
    def grid(self):
        return self._env.get_map_with_agents()

    def get_state(self):
        """agent.py:76-78: the agent's window of the map with agents and beams (characters)."""
        return return_view(self._env.get_map_with_agents(), self.pos, self.row_size, self.col_size)

    def action_map(self, action_number):
        return self.action_table[action_number]

    def get_done(self):
        return False


class MapEnv(MultiAgentEnv):
    GAME = K.GAME_HARVEST
    agent_class = Agent

    def __init__(self, ascii_map, num_agents=1, render=True, color_map=None, return_agent_actions=False,
                 seed=None, device=0, view_len=K.VIEW_LEN, env_index=0):
        self.num_agents = num_agents
        self.base_map = self.ascii_to_numpy(ascii_map)
        self.return_agent_actions = return_agent_actions
        self.view_len = view_len
        self.color_map = dict(DEFAULT_COLOURS if color_map is None else color_map)
        self.color_map.update({k: list(v) for k, v in K.CLEANUP_COLOURS.items() if color_map is None})
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1)) * 2**31 + int(np.random.randint(0, 2**31 - 1))
        self.seed = seed
        self._engine = VecEngine(self.GAME, [str(r) for r in ascii_map], num_envs=1, num_agents=num_agents,
                                 view_len=view_len, seed=seed, env_index_base=env_index, device=device,
                                 keep_beams=True, color_map=self.color_map)
        self.spawn_points = [[r, c] for r in range(self.base_map.shape[0]) for c in range(self.base_map.shape[1])
                             if self.base_map[r, c] == 'P']
        self.wall_points = [[r, c] for r in range(self.base_map.shape[0]) for c in range(self.base_map.shape[1])
                            if self.base_map[r, c] == '@']
        self.agents = {}
        self._cache = None
        self.setup_agents()
        self._engine.reset_host()
        self._dirty()

    # ------------------------------------------------------------------ plumbing
    def _state(self):
        if self._cache is None:
            self._cache = self._engine.get_state()
        return self._cache

    def _dirty(self):
        self._cache = None

    def setup_agents(self):
        """harvest.py:46-55 / cleanup.py:118-130: the engine spawns positions and rotations in reset();
        this only (re)builds the Python-side views."""
        self.agents = {}
        for i in range(self.num_agents):
            agent = self.agent_class(self, i)
            self.agents[agent.agent_id] = agent

    def ascii_to_numpy(self, ascii_list):
        """The ASCII rows as a '<U1' character grid (what map_env.py:132-150 builds cell by cell)."""
        return np.array([list(line) for line in ascii_list], dtype='<U1')

    @staticmethod
    def _to_chars(grid_i8):
        return grid_i8.view(np.uint8).astype(np.uint32).view('<U1').reshape(grid_i8.shape)

    @staticmethod
    def _to_i8(chars):
        chars = np.asarray(chars)
        return np.array([[ord(ch) for ch in row] for row in chars], dtype=np.int8)

    # ------------------------------------------------------------------ the MultiAgentEnv API
    def _wrap_obs(self, obs_u8, actions):
        observations = {}
        ids = list(self.agents.keys())
        for i, agent_id in enumerate(ids):
            rgb_arr = _NORMALISE[obs_u8[0, i]]               # == (rgb - 128.0) / 255.0, float64
            if self.return_agent_actions:                    # map_env.py:201-205 / :242-246
                if actions is None:
                    prev_actions = np.array([0 for _ in range(self.num_agents - 1)]).astype(np.int64)
                else:
                    prev_actions = np.array([actions[key] for key in sorted(actions.keys())
                                             if key != agent_id]).astype(np.int64)
                observations[agent_id] = {"curr_obs": rgb_arr, "other_agent_actions": prev_actions,
                                          "visible_agents": self.find_visible_agents(agent_id)}
            else:
                observations[agent_id] = rgb_arr
        return observations

    def step(self, actions):
        """map_env.py:152-212.  actions: {agent_id: int}; any subset of the agents, in any order."""
        N = self.num_agents
        act = np.full((1, N), K.NO_ACTION, dtype=np.int32)
        order = np.full((1, N), 0xFF, dtype=np.uint8)
        for k, (agent_id, action) in enumerate(actions.items()):
            agent = self.agents[agent_id]                    # KeyError for an unknown agent, as in the reference
            agent.action_map(action)                         # KeyError for an action outside the game's table
            act[0, agent._i] = int(action)
            order[0, k] = agent._i
        obs, rew, done = self._engine.step_host(act, order)
        self._dirty()
        observations = self._wrap_obs(obs, actions)
        rewards, dones = {}, {}
        for i, agent_id in enumerate(self.agents.keys()):
            rewards[agent_id] = int(rew[0, i])
            dones[agent_id] = bool(done[0, i])
        dones["__all__"] = np.any(list(dones.values()))
        return observations, rewards, dones, {}

    def reset(self):
        """map_env.py:214-249 (observations are not rotated in reset)."""
        self.setup_agents()
        obs = self._engine.reset_host()
        self._dirty()
        return self._wrap_obs(obs, None)

    # ------------------------------------------------------------------ state views the reference exposes
    @property
    def world_map(self):
        return self._to_chars(self._state()["world"][0])

    @world_map.setter
    def world_map(self, grid):
        self._engine.set_state(world=self._to_i8(grid)[None])
        self._dirty()

    @property
    def beam_pos(self):
        """[(row, col, char)] of the beams drawn by the last step (map_env.py:86,648); order within a step is not kept."""
        beam = self._state()["beam"][0]
        return [(int(r), int(c), chr(int(beam[r, c]))) for r, c in zip(*np.nonzero(beam))]

    @property
    def agent_pos(self):
        return [p.tolist() for p in self._state()["pos"][0].astype(np.int64)]

    @property
    def test_map(self):
        """map_env.py:257-278: world + 'P' for agents + beams."""
        grid = self.world_map.copy()
        for r, c in self.agent_pos:
            grid[r, c] = 'P'
        for r, c, ch in self.beam_pos:
            grid[r, c] = ch
        return grid

    def get_map_with_agents(self):
        """map_env.py:280-302: world + agent glyphs (index order) + beams."""
        grid = self.world_map.copy()
        for i, (r, c) in enumerate(self.agent_pos):
            grid[r, c] = str(int(str(i)[-1]) + 1)[0]         # '<U1' truncation: agent-9 renders as '1'
        for r, c, ch in self.beam_pos:
            grid[r, c] = ch
        return grid

    def update_map(self, new_points):
        """map_env.py:554-558."""
        world = self._state()["world"].copy()
        for row, col, char in new_points:
            world[0, row, col] = ord(char)
        self._engine.set_state(world=world)
        self._dirty()

    def map_to_colors(self, map=None, color_map=None):
        """map_env.py:316-339.  With no arguments the full frame is rendered on the GPU."""
        if map is None and color_map is None:
            return self._engine.render_full(0).astype(int)
        if map is None:
            map = self.get_map_with_agents()
        if color_map is None:
            color_map = self.color_map
        # one table lookup for the whole grid: glyph code point -> RGB row (a glyph without a colour is a KeyError, as it is
        # when the reference indexes its dict cell by cell)
        glyphs = np.asarray(map, dtype='<U1')
        codes = np.ascontiguousarray(glyphs).view(np.uint32).reshape(glyphs.shape)
        # (an empty cell of a '<U1' array is code point 0 and reads back as '': the reference's DEFAULT_COLOURS has that key,
        # map_env.py:26; keys that are not one character -- '' aside -- can never equal a cell and are skipped)
        def glyph_of(code):
            return chr(int(code)) if code else ''
        for code in np.unique(codes):
            if glyph_of(code) not in color_map:
                raise KeyError(glyph_of(code))
        table = np.zeros((int(codes.max()) + 1 if codes.size else 1, 3), dtype=int)
        for glyph, rgb in color_map.items():
            if not isinstance(glyph, str) or len(glyph) > 1:
                continue
            code = ord(glyph) if glyph else 0
            if code < table.shape[0]:
                table[code] = rgb
        return table[codes]

    def render(self, filename=None):
        """map_env.py:341-355."""
        import matplotlib.pyplot as plt
        plt.imshow(self.map_to_colors(), interpolation='nearest')
        if filename is None:
            plt.show()
        else:
            plt.savefig(filename)

    def rotate_view(self, orientation, view):
        """Quarter turns of the egocentric view by orientation (map_env.py:669-689): UP 0, LEFT 1, DOWN 2, RIGHT 3."""
        quarter_turns = {'UP': 0, 'LEFT': 1, 'DOWN': 2, 'RIGHT': 3}
        if orientation not in quarter_turns:
            raise ValueError('Orientation {} is not valid'.format(orientation))
        k = quarter_turns[orientation]
        return view if k == 0 else np.rot90(view, k)

    def find_visible_agents(self, agent_id):
        """map_env.py:749-770.  The reference compares the agent's OWN position with its window for every
        other agent (:767), so the result is all ones; reproduced as is."""
        return np.array([1 for other in sorted(self.agents.keys()) if other != agent_id])

    def test_if_in_bounds(self, pos):
        return 0 <= pos[0] < self.base_map.shape[0] and 0 <= pos[1] < self.base_map.shape[1]

    # ------------------------------------------------------------------ spaces (harvest.py:30-44, cleanup.py:68-82)
    @property
    def observation_space(self):
        shape = (2 * self.view_len + 1, 2 * self.view_len + 1, 3)
        if self.return_agent_actions:
            return Dict({"curr_obs": Box(low=-np.inf, high=np.inf, shape=shape, dtype=np.float32),
                         "other_agent_actions": Box(low=0, high=len(ACTIONS), shape=(self.num_agents - 1,), dtype=np.int32),
                         "visible_agents": Box(low=0, high=self.num_agents, shape=(self.num_agents - 1,), dtype=np.int32)})
        return Box(low=0.0, high=0.0, shape=shape, dtype=np.float32)

    def close(self):
        self._engine.close()
