"""Host-side derivation of the engine configuration from an ASCII map: what the reference
constructors compute (map_env.py:62-102, harvest.py:20-28, cleanup.py:32-66) expressed as
the tables the kernels consume."""
import numpy as np

from . import constants as K
from . import prng


def ascii_to_bytes(ascii_map):
    """ascii_to_numpy (map_env.py:132-150) as one row-major byte string."""
    rows = [str(r) for r in ascii_map]
    if not rows or any(len(r) != len(rows[0]) for r in rows):
        raise ValueError("ascii_map must be a non-empty list of equal-length strings")
    flat = "".join(rows).encode("ascii")
    return flat, len(rows), len(rows[0])


def make_lut(color_map=None):
    """128 x 3 uint8 glyph -> RGB table.  Default = DEFAULT_COLOURS updated with CLEANUP_COLORS,
    which is what both games see once a CleanupEnv has been built (cleanup.py:64 mutates the
    module-global dict)."""
    lut = np.zeros((128, 3), dtype=np.uint8)
    tables = (K.DEFAULT_COLOURS, K.CLEANUP_COLOURS) if color_map is None else (color_map,)
    for table in tables:
        for ch, rgb in table.items():
            if len(ch) == 1 and ord(ch) < 128:
                lut[ord(ch)] = rgb
    return lut


def harvest_thresholds():
    """rand < SPAWN_PROB[min(n, 3)] (harvest.py:13,100-102) as integer thresholds."""
    return np.array([prng.threshold(p) for p in K.HARVEST_SPAWN_PROB], dtype=np.uint64)


def cleanup_probabilities(potential_waste_area, n_waste):
    """compute_probabilities (cleanup.py:156-171) with compute_permitted_area (:173-179),
    evaluated with the same Python float operations as the reference.
    Returns (current_apple_spawn_prob, current_waste_spawn_prob)."""
    waste_density = 0
    if potential_waste_area > 0:
        waste_density = 1 - (potential_waste_area - n_waste) / potential_waste_area
    if waste_density >= K.THRESHOLD_DEPLETION:
        return 0, 0
    if waste_density <= K.THRESHOLD_RESTORATION:
        return K.APPLE_RESPAWN_PROBABILITY, K.WASTE_SPAWN_PROBABILITY
    spawn_prob = (1 - (waste_density - K.THRESHOLD_RESTORATION)
                  / (K.THRESHOLD_DEPLETION - K.THRESHOLD_RESTORATION)) * K.APPLE_RESPAWN_PROBABILITY
    return spawn_prob, K.WASTE_SPAWN_PROBABILITY


def cleanup_thresholds(potential_waste_area):
    """Threshold tables indexed by the current number of 'H' cells (0..potential_waste_area)."""
    ta = np.zeros(potential_waste_area + 1, dtype=np.uint64)
    tw = np.zeros(potential_waste_area + 1, dtype=np.uint64)
    for n in range(potential_waste_area + 1):
        pa, pw = cleanup_probabilities(potential_waste_area, n)
        ta[n], tw[n] = prng.threshold(pa), prng.threshold(pw)
    return ta, tw


def potential_waste_area(ascii_map):
    """cleanup.py:36-38: number of 'H' plus 'R' cells of the base map."""
    return sum(r.count('H') + r.count('R') for r in ascii_map)


def algorithmic_bytes_per_env_step(H, W, N, V=15):
    """grid read + write-back, uint8 obs, actions, rewards, dones, agent state, header."""
    return 2 * H * W + N * V * V * 3 + N * 4 + N * 4 + N * 1 + N * 16 + 8
