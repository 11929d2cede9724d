"""HarvestEnv on the MI355X engine (reference: social_dilemmas/envs/harvest.py:18)."""
import numpy as np

from . import constants as K
from .map_env import Agent, Discrete, MapEnv

APPLE_RADIUS = K.APPLE_RADIUS
SPAWN_PROB = list(K.HARVEST_SPAWN_PROB)
HARVEST_VIEW_SIZE = K.VIEW_LEN
HARVEST_ACTIONS = dict(K.HARVEST_ACTIONS)


class HarvestAgent(Agent):
    """agent.py:152-183: FIRE costs 1, being hit by 'F' costs 50, an apple pays 1 (applied by the kernel)."""
    action_table = HARVEST_ACTIONS


class HarvestEnv(MapEnv):
    GAME = K.GAME_HARVEST
    agent_class = HarvestAgent

    def __init__(self, ascii_map=K.HARVEST_MAP, num_agents=1, render=False, return_agent_actions=False, **engine_kw):
        super().__init__(ascii_map, num_agents, render, return_agent_actions=return_agent_actions, **engine_kw)
        self.apple_points = [[r, c] for r in range(self.base_map.shape[0]) for c in range(self.base_map.shape[1])
                             if self.base_map[r, c] == 'A']

    @property
    def action_space(self):
        return Discrete(8)

    def count_apples(self, window):
        """harvest.py:106-111."""
        unique, counts = np.unique(window, return_counts=True)
        return dict(zip(unique, counts)).get('A', 0)
