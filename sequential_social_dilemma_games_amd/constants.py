"""Game data of the Harvest / Cleanup gridworlds: maps, glyph colours, action and
orientation tables.  These are the *data* the reference engine is defined on
(`social_dilemmas/constants.py:7-50`, `map_env.py:11-41`, `agent.py:7-13,148-149,186-188`,
`cleanup.py:15-27`, `harvest.py:8-15`); the engine's kernels take them as configuration.

Cell alphabet of a world grid: ' ' empty, '@' wall, 'A' apple, 'H' waste, 'R' river,
'S' stream.  Base maps additionally use 'P' (agent spawn point) and, in Cleanup,
'B' (apple spawn point).  View-only glyphs: '1'..'9' agents, 'F'/'C' beams, '0' padding.
"""

import os

_MAPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps")


def load_map(name):
    """ASCII map shipped as a data file under maps/ (one row per line)."""
    with open(os.path.join(_MAPS, name + ".txt")) as f:
        rows = [line.rstrip("\n") for line in f if line.strip("\n")]
    if not rows or any(len(r) != len(rows[0]) for r in rows):
        raise ValueError("map %s is not rectangular" % name)
    return rows


# The reference's two maps (data, social_dilemmas/constants.py:7-50): Harvest is 16 rows x 38 cols -- BASELINE.json
# labels it "25x38", but no such map exists in the reference -- and Cleanup is 25 rows x 18 cols.
HARVEST_MAP = load_map("harvest_16x38")
CLEANUP_MAP = load_map("cleanup_25x18")


def harvest_map_25x38():
    """Synthetic 25 x 38 Harvest map honouring BASELINE.json's label:
    rows[0:15] + rows[6:15] + rows[15:16] of HARVEST_MAP (wall-closed, 31 'P', 252 'A')."""
    m = HARVEST_MAP
    return m[0:15] + m[6:15] + m[15:16]


def cleanup_map_48x36():
    """Synthetic 48 x 36 Cleanup map for config C5: wall + 2 x interior
    rows 1..23 + wall; each interior row '@' + r[1:17] + '  ' + r[1:17] + '@'."""
    m = CLEANUP_MAP
    wall = '@' * 36
    interior = ['@' + r[1:17] + '  ' + r[1:17] + '@' for r in m[1:24]]
    return [wall] + interior + interior + [wall]


# Action ids (agent.py:7-13; 7 = FIRE in both games, 8 = CLEAN in Cleanup only).
MOVE_LEFT, MOVE_RIGHT, MOVE_UP, MOVE_DOWN, STAY, TURN_CLOCKWISE, TURN_COUNTERCLOCKWISE, FIRE, CLEAN = range(9)
NO_ACTION = -1

BASE_ACTIONS = {0: 'MOVE_LEFT', 1: 'MOVE_RIGHT', 2: 'MOVE_UP', 3: 'MOVE_DOWN', 4: 'STAY',
                5: 'TURN_CLOCKWISE', 6: 'TURN_COUNTERCLOCKWISE'}
HARVEST_ACTIONS = dict(list(BASE_ACTIONS.items()) + [(7, 'FIRE')])
CLEANUP_ACTIONS = dict(list(BASE_ACTIONS.items()) + [(7, 'FIRE'), (8, 'CLEAN')])

# Orientation codes follow the key order of the reference's ORIENTATIONS dict
# (map_env.py:19-22) because spawn_rotation indexes that order (map_env.py:664-667).
ORIENTATION_NAMES = ('LEFT', 'RIGHT', 'UP', 'DOWN')
ORIENTATION_CODE = {n: i for i, n in enumerate(ORIENTATION_NAMES)}
ORIENTATION_VEC = ((-1, 0), (1, 0), (0, -1), (0, 1))   # (d_row, d_col), also the MOVE_* vectors

# Glyph -> RGB (map_env.py:24-41; cleanup.py:15-18).
DEFAULT_COLOURS = {' ': (0, 0, 0), '0': (0, 0, 0), '@': (180, 180, 180), 'A': (0, 255, 0),
                   'F': (255, 255, 0), 'P': (159, 67, 255),
                   '1': (159, 67, 255), '2': (2, 81, 154), '3': (204, 0, 204), '4': (216, 30, 54),
                   '5': (254, 151, 0), '6': (100, 255, 255), '7': (99, 99, 255),
                   '8': (250, 204, 255), '9': (238, 223, 16)}
CLEANUP_COLOURS = {'C': (100, 255, 255), 'S': (113, 75, 24), 'H': (99, 156, 194), 'R': (113, 75, 24)}

GAME_HARVEST = 0
GAME_CLEANUP = 1

# Module-level rule constants of the reference (harvest.py:8-15, cleanup.py:20-27).
APPLE_RADIUS = 2            # with the j*j + k*k <= 2 test this selects the 3x3 neighbourhood
HARVEST_SPAWN_PROB = (0.0, 0.005, 0.02, 0.05)
VIEW_LEN = 7                # HARVEST_VIEW_SIZE == CLEANUP_VIEW_SIZE
BEAM_LEN = 5                # ACTIONS['FIRE'] == ACTIONS['CLEAN']
THRESHOLD_DEPLETION = 0.4
THRESHOLD_RESTORATION = 0.0
WASTE_SPAWN_PROBABILITY = 0.5
APPLE_RESPAWN_PROBABILITY = 0.05
REWARD_APPLE = 1
REWARD_FIRE = -1
REWARD_HIT = -50
