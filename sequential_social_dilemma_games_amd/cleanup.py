"""CleanupEnv on the MI355X engine (reference: social_dilemmas/envs/cleanup.py:30)."""
import numpy as np

from . import config
from . import constants as K
from .map_env import Agent, Discrete, MapEnv

CLEANUP_COLORS = {k: list(v) for k, v in K.CLEANUP_COLOURS.items()}
CLEANUP_VIEW_SIZE = K.VIEW_LEN
CLEANUP_ACTIONS = dict(K.CLEANUP_ACTIONS)
thresholdDepletion = K.THRESHOLD_DEPLETION
thresholdRestoration = K.THRESHOLD_RESTORATION
wasteSpawnProbability = K.WASTE_SPAWN_PROBABILITY
appleRespawnProbability = K.APPLE_RESPAWN_PROBABILITY


class CleanupAgent(Agent):
    """agent.py:191-222: FIRE costs 1, CLEAN is free, 'F' hits cost 50, 'C' hits nothing."""
    action_table = CLEANUP_ACTIONS


class CleanupEnv(MapEnv):
    GAME = K.GAME_CLEANUP
    agent_class = CleanupAgent

    def __init__(self, ascii_map=K.CLEANUP_MAP, num_agents=1, render=False, return_agent_actions=False, **engine_kw):
        self._probs = None
        super().__init__(ascii_map, num_agents, render, return_agent_actions=return_agent_actions, **engine_kw)
        bm = self.base_map
        cells = [(r, c) for r in range(bm.shape[0]) for c in range(bm.shape[1])]
        self.potential_waste_area = int(np.sum(bm == 'H') + np.sum(bm == 'R'))      # cleanup.py:36-38
        self.apple_points = [[r, c] for r, c in cells if bm[r, c] == 'B']
        self.waste_start_points = [[r, c] for r, c in cells if bm[r, c] == 'H']
        self.waste_points = [[r, c] for r, c in cells if bm[r, c] in 'HR']
        self.river_points = [[r, c] for r, c in cells if bm[r, c] == 'R']
        self.stream_points = [[r, c] for r, c in cells if bm[r, c] == 'S']
        assert self.potential_waste_area == self._engine.potential_waste_area

    @property
    def action_space(self):
        return Discrete(9)

    def compute_permitted_area(self):
        """cleanup.py:173-179."""
        return self.potential_waste_area - int(np.sum(self.world_map == 'H'))

    def compute_probabilities(self):
        """cleanup.py:156-171 on the current map: updates and returns (current_apple_spawn_prob, current_waste_spawn_prob)."""
        n_waste = self.potential_waste_area - self.compute_permitted_area()
        self._probs = config.cleanup_probabilities(self.potential_waste_area, n_waste)
        return self._probs

    def _probs_of_last_step(self):
        # The reference's attributes hold what custom_map_update() computed inside the last step / reset,
        # after the beams and before the spawn (cleanup.py:113-116); the kernel records the waste count it used.
        if self._probs is None:
            self._probs = config.cleanup_probabilities(self.potential_waste_area, int(self._engine.waste_count()[0]))
        return self._probs

    def _dirty(self):
        super()._dirty()
        self._probs = None

    @property
    def current_apple_spawn_prob(self):
        return self._probs_of_last_step()[0]

    @property
    def current_waste_spawn_prob(self):
        return self._probs_of_last_step()[1]
