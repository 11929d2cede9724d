"""Shared counter-based PRNG of the engine (host mirror).

The reference draws from two global Mersenne Twisters by *sequence position*
(`map_env.py:422`, `harvest.py:101`, `cleanup.py:139,145,150`,
`map_env.py:656,666`), which no parallel kernel can reproduce.  The engine
instead keys every draw on *what it is for*:

    draw = H(seed, env, episode, t, stream, index)

`H` is a chain of one 32-bit bijective mixer ("triple32", Wellons) absorbing
one word at a time: ``h <- mix32(h ^ word)``.  The (seed, env, episode) prefix
is folded once per reset into a per-env key; (t, stream) once per phase; only
the last `mix32` is paid per draw.  The HIP kernels (csrc/ssd_device.hpp), the
C oracle (oracle/ssd_oracle.c) and this file are three independent statements
of the same function; tests/test_prng.py pins them against each other.

Streams (the `index` word is given in brackets):
    SPAWN_POINT [agent << 16 | cell]  reset, t = 0   (replaces map_env.py:656)
    SPAWN_ROT   [agent]               reset, t = 0   (replaces map_env.py:666)
    MOVE        [i]   Fisher-Yates position i = n-1..1 (replaces map_env.py:422)
    APPLE       [cell]                (replaces harvest.py:101, cleanup.py:139)
    WASTE_COIN  [cell]                (replaces cleanup.py:150)
    WASTE_ORDER [cell]                (replaces cleanup.py:145)
    ACTION      [agent]               random-action rollouts (rollout.py:65)
`cell` is `row * W + col`; `t` is 0 for reset and k for the k-th step after it.
"""

import numpy as np

M32 = 0xFFFFFFFF
H0 = 0x243F6A88

S_SPAWN_POINT = 1
S_SPAWN_ROT = 2
S_MOVE = 3
S_APPLE = 4
S_WASTE_COIN = 5
S_WASTE_ORDER = 6
S_ACTION = 7


def mix32(x):
    """triple32: 32-bit bijection with low avalanche bias (scalar ints)."""
    x &= M32
    x ^= x >> 17
    x = (x * 0xED5AD4BB) & M32
    x ^= x >> 11
    x = (x * 0xAC4C1B51) & M32
    x ^= x >> 15
    x = (x * 0x31848BAB) & M32
    x ^= x >> 14
    return x


def env_key(seed, env, episode):
    """Per-env, per-episode key: absorbs seed_lo, seed_hi, env, episode."""
    h = H0
    for w in (seed & M32, (seed >> 32) & M32, env & M32, episode & M32):
        h = mix32(h ^ w)
    return h


def phase_key(key, t, stream):
    return mix32(mix32(key ^ (t & M32)) ^ stream)


def draw(pkey, index):
    """One uniform u32."""
    return mix32(pkey ^ (index & M32))


def draw_full(seed, env, episode, t, stream, index):
    return draw(phase_key(env_key(seed, env, episode), t, stream), index)


def randint(u32, n):
    """Uniform integer in [0, n) from one u32 (multiply-shift, no rejection)."""
    return (u32 * n) >> 32


def threshold(p):
    """Smallest T with (k / 2**32 < p)  <=>  (k < T) for every u32 k.

    `p * 2**32` is exact in float64 (power-of-two scaling), so the ceiling is
    exact too.  p <= 0 -> 0 (never), p >= 1 -> 2**32 (always; callers clamp).
    """
    import math
    if p <= 0.0:
        return 0
    return min(int(math.ceil(p * 4294967296.0)), 1 << 32)


# ---- vectorised forms (numpy uint32 arrays), used by host-side action generation ----

def mix32_np(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(17)
    x = (x * np.uint64(0xED5AD4BB)) & M32
    x ^= x >> np.uint64(11)
    x = (x * np.uint64(0xAC4C1B51)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x31848BAB)) & M32
    x ^= x >> np.uint64(14)
    return x


def random_actions(seed, env_ids, episodes, t, num_agents, num_actions):
    """Actions the ACTION stream yields for step `t` (same as the device kernel).

    env_ids, episodes: int arrays [E].  Returns int32 [E, num_agents].
    """
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    episodes = np.asarray(episodes, dtype=np.uint64)
    h = np.full(env_ids.shape, H0, dtype=np.uint64)
    h = mix32_np(h ^ np.uint64(seed & M32))
    h = mix32_np(h ^ np.uint64((seed >> 32) & M32))
    h = mix32_np(h ^ (env_ids & M32))
    h = mix32_np(h ^ (episodes & M32))
    pk = mix32_np(mix32_np(h ^ np.uint64(t & M32)) ^ np.uint64(S_ACTION))
    idx = np.arange(num_agents, dtype=np.uint64)[None, :]
    u = mix32_np(pk[:, None] ^ idx)
    return ((u * np.uint64(num_actions)) >> np.uint64(32)).astype(np.int32)
