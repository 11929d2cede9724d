"""MI355X-native vectorised engine for the Harvest / Cleanup social-dilemma gridworlds.

    from sequential_social_dilemma_games_amd import HarvestEnv, CleanupEnv     # dict API (RLlib MultiAgentEnv)
    from sequential_social_dilemma_games_amd import VecEngine                  # batched tensor API

Everything that steps an env goes through libssd_hip.so (include/ssd.h); importing this package does
not load it, constructing an env does -- and fails loudly if it is missing.
"""
from .constants import CLEANUP_MAP, HARVEST_MAP  # noqa: F401


def __getattr__(name):
    if name == "VecEngine":
        from .engine import VecEngine
        return VecEngine
    if name in ("HarvestEnv", "HarvestAgent"):
        from . import harvest
        return getattr(harvest, name)
    if name in ("CleanupEnv", "CleanupAgent"):
        from . import cleanup
        return getattr(cleanup, name)
    if name == "MapEnv":
        from .map_env import MapEnv
        return MapEnv
    raise AttributeError(name)
