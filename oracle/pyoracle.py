"""ctypes wrapper of oracle/libssd_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (as the checker).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    so = os.path.join(_HERE, "libssd_oracle.so")
    src = os.path.join(_HERE, "ssd_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libssd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libssd_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp = C.c_void_p
        L.ssd_oracle_create.restype = vp
        L.ssd_oracle_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_uint64, C.c_uint32, vp]
        L.ssd_oracle_destroy.argtypes = [vp]
        L.ssd_oracle_reset.argtypes = [vp, vp, vp]
        L.ssd_oracle_step.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ssd_oracle_step_random.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.ssd_oracle_get_state.argtypes = [vp] + [vp] * 6
        L.ssd_oracle_set_state.argtypes = [vp] + [vp] * 6
        L.ssd_oracle_observe.argtypes = [vp, C.c_int, vp]
        L.ssd_oracle_cleanup_thresholds.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ssd_oracle_potential_waste_area.argtypes = [vp]
        L.ssd_oracle_draw.restype = C.c_uint32
        L.ssd_oracle_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _LIB = L
    return _LIB


def make_lut(colours):
    lut = np.zeros((128, 3), dtype=np.uint8)
    for ch, rgb in colours.items():
        if len(ch) == 1:
            lut[ord(ch)] = rgb
    return lut


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Batched CPU env: state in host arrays, one serial loop over envs per call."""

    def __init__(self, game, ascii_map, num_envs, num_agents, lut, view_len=7, beam_len=5, seed=0, env_base=0):
        self.game, self.E, self.N = int(game), int(num_envs), int(num_agents)
        self.H, self.W = len(ascii_map), len(ascii_map[0])
        self.view_len, self.V = view_len, 2 * view_len + 1
        self.num_actions = 8 if game == 0 else 9
        flat = "".join(ascii_map).encode("ascii")
        assert len(flat) == self.H * self.W
        self._lut = np.ascontiguousarray(lut, dtype=np.uint8)
        self._h = lib().ssd_oracle_create(self.game, self.H, self.W, flat, self.E, self.N, view_len, beam_len,
                                          seed, env_base, _p(self._lut))
        if not self._h:
            raise ValueError("ssd_oracle_create rejected the configuration (open map? too many agents?)")

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ssd_oracle_destroy(self._h)
            except TypeError:                      # (interpreter shutdown: the module's globals are already gone)
                pass
            self._h = None

    def _obs_buf(self):
        return np.zeros((self.E, self.N, self.V, self.V, 3), dtype=np.uint8)

    def reset(self, mask=None):
        obs = self._obs_buf()
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib().ssd_oracle_reset(self._h, _p(m), _p(obs))
        if rc:
            raise RuntimeError("ssd_oracle_reset failed: %d" % rc)
        return obs

    def step(self, actions, order=None):
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint8).reshape(self.E, self.N)
        obs = self._obs_buf()
        rew = np.zeros((self.E, self.N), dtype=np.int32)
        done = np.zeros((self.E, self.N), dtype=np.uint8)
        rc = lib().ssd_oracle_step(self._h, _p(actions), _p(order), _p(obs), _p(rew), _p(done))
        if rc:
            raise RuntimeError("ssd_oracle_step failed: %d" % rc)
        return obs, rew, done

    def step_random(self, want_obs=True):
        act = np.zeros((self.E, self.N), dtype=np.int32)
        obs = self._obs_buf() if want_obs else None
        rew = np.zeros((self.E, self.N), dtype=np.int32)
        done = np.zeros((self.E, self.N), dtype=np.uint8)
        rc = lib().ssd_oracle_step_random(self._h, self.num_actions, _p(act), _p(obs), _p(rew), _p(done))
        if rc:
            raise RuntimeError("ssd_oracle_step_random failed: %d" % rc)
        return act, obs, rew, done

    def get_state(self):
        E, N, H, W = self.E, self.N, self.H, self.W
        s = dict(world=np.zeros((E, H, W), np.int8), beam=np.zeros((E, H, W), np.int8),
                 pos=np.zeros((E, N, 2), np.int16), orient=np.zeros((E, N), np.uint8),
                 episode=np.zeros(E, np.uint32), t=np.zeros(E, np.uint32))
        lib().ssd_oracle_get_state(self._h, _p(s["world"]), _p(s["beam"]), _p(s["pos"]), _p(s["orient"]),
                                   _p(s["episode"]), _p(s["t"]))
        return s

    def set_state(self, world=None, beam=None, pos=None, orient=None, episode=None, t=None):
        E, N, H, W = self.E, self.N, self.H, self.W

        def prep(a, dt, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape == shape, (a.shape, shape)
            return a
        world, beam = prep(world, np.int8, (E, H, W)), prep(beam, np.int8, (E, H, W))
        pos, orient = prep(pos, np.int16, (E, N, 2)), prep(orient, np.uint8, (E, N))
        episode, t = prep(episode, np.uint32, (E,)), prep(t, np.uint32, (E,))
        lib().ssd_oracle_set_state(self._h, _p(world), _p(beam), _p(pos), _p(orient), _p(episode), _p(t))

    def observe(self, rotate=True):
        obs = self._obs_buf()
        lib().ssd_oracle_observe(self._h, int(bool(rotate)), _p(obs))
        return obs

    def cleanup_thresholds(self, n_waste):
        a, w = C.c_uint64(), C.c_uint64()
        lib().ssd_oracle_cleanup_thresholds(self._h, int(n_waste), C.byref(a), C.byref(w))
        return a.value, w.value

    @property
    def potential_waste_area(self):
        return lib().ssd_oracle_potential_waste_area(self._h)


def draw(seed, env, episode, t, stream, index):
    return lib().ssd_oracle_draw(seed, env, episode, t, stream, index)
