"""Batched engine: E independent Harvest / Cleanup envs resident in the HBM of one MI355X.

This is the throughput API behind the dict-API env classes (map_env.py): actions i32 [E,N] in,
uint8 observations [E,N,V,V,3], i32 rewards and u8 dones out, one fused HIP kernel launch per
step (csrc/ssd_kernels.hip) reached through the C ABI of include/ssd.h.

Two calling styles:
  * device tensors (`reset`, `step`, `step_random`, `observe`): torch is used only to own the
    output buffers and to name the HIP stream; calls are asynchronous on torch's current stream;
  * host arrays (`*_host`): numpy in / numpy out, the library stages through its own device
    buffers and returns when the results have landed.
"""
import ctypes as C

import numpy as np

from . import _capi
from . import config as cfgmod
from . import constants as K


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class VecEngine(object):
    def __init__(self, game, ascii_map=None, num_envs=1, num_agents=1, view_len=K.VIEW_LEN, beam_len=K.BEAM_LEN,
                 seed=0, env_index_base=0, device=0, keep_beams=False, color_map=None):
        self.game = int(game)
        if ascii_map is None:
            ascii_map = K.HARVEST_MAP if self.game == K.GAME_HARVEST else K.CLEANUP_MAP
        self.ascii_map = [str(r) for r in ascii_map]
        flat, self.H, self.W = cfgmod.ascii_to_bytes(self.ascii_map)
        self.E, self.N = int(num_envs), int(num_agents)
        self.view_len, self.V, self.beam_len = int(view_len), 2 * int(view_len) + 1, int(beam_len)
        self.seed, self.env_index_base, self.device = int(seed), int(env_index_base), int(device)
        self.keep_beams = bool(keep_beams)
        self.num_actions = 8 if self.game == K.GAME_HARVEST else 9     # harvest.py:44, cleanup.py:70
        self._lut = np.ascontiguousarray(cfgmod.make_lut(color_map))
        self._thr_h = cfgmod.harvest_thresholds()
        self.potential_waste_area = cfgmod.potential_waste_area(self.ascii_map) if self.game == K.GAME_CLEANUP else 0
        self._thr_ca, self._thr_cw = cfgmod.cleanup_thresholds(self.potential_waste_area)
        self._flat = flat
        c = _capi.SsdConfig()
        c.struct_size = C.sizeof(_capi.SsdConfig)
        c.game, c.height, c.width, c.base_map = self.game, self.H, self.W, flat
        c.num_envs, c.num_agents, c.view_len, c.beam_len = self.E, self.N, self.view_len, self.beam_len
        c.seed, c.env_index_base, c.device_id, c.keep_beams = self.seed, self.env_index_base, self.device, int(self.keep_beams)
        c.color_lut = self._lut.ctypes.data
        c.harvest_thresholds = self._thr_h.ctypes.data
        c.cleanup_apple_thresholds = self._thr_ca.ctypes.data
        c.cleanup_waste_thresholds = self._thr_cw.ctypes.data
        self._h = C.c_void_p()
        L = _capi.lib()
        _capi.check(L.ssd_create(C.byref(c), C.byref(self._h)))
        self._L = L
        self._out_cache = (None, None)               # (outputs tuple, its pointers) of the last step_random call
        # Steps since the last reset of ALL envs, or None once envs may be at different points of their episodes (a masked
        # reset, set_state(t=...)): lets SSDVectorEnv know, without asking the device, on which step everybody reaches the
        # horizon.
        self.steps_since_full_reset = None
        self._torch_dev = None
        if L.ssd_potential_waste_area(self._h) != self.potential_waste_area:
            raise _capi.SsdError("potential_waste_area mismatch between host and library")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.ssd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ device-tensor API
    def _torch(self):
        if self._torch_dev is None:
            import torch
            self._torch_dev = (torch, torch.device("cuda", self.device))
        return self._torch_dev

    def _stream(self):
        """torch's current stream on the engine's device (what the kernels are enqueued on)."""
        torch, dev = self._torch()
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the cheap way; the public API builds a Stream object
        if raw is not None:
            return C.c_void_p(raw(dev.index if dev.index is not None else torch.cuda.current_device()))
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def alloc_outputs(self, float32=False):
        """(obs u8 -- or float32 -- [E,N,V,V,3], rew i32 [E,N], done u8 [E,N]) on the engine's device."""
        torch, dev = self._torch()
        return (torch.empty((self.E, self.N, self.V, self.V, 3), dtype=torch.float32 if float32 else torch.uint8, device=dev),
                torch.empty((self.E, self.N), dtype=torch.int32, device=dev),
                torch.empty((self.E, self.N), dtype=torch.uint8, device=dev))

    @staticmethod
    def _dp(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _check_tensor(self, t, shape, dtype, name):
        torch, dev = self._torch()
        if t.device != dev or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), dev))

    def _obs_flags(self, obs):
        """Observation buffers may be uint8 (default) or float32: float32((u8 - 128.0) / 255.0), written by the
        kernel itself (SSD_OBS_F32) -- what the first layer of a policy network consumes."""
        torch, dev = self._torch()
        if obs is None:
            return 0
        if obs.dtype not in (torch.uint8, torch.float32):
            raise ValueError("obs must be uint8 or float32")
        self._check_tensor(obs, (self.E, self.N, self.V, self.V, 3), obs.dtype, "obs")
        return _capi.SSD_OBS_F32 if obs.dtype == torch.float32 else 0

    def reset(self, mask=None, obs=None):
        """MapEnv.reset (map_env.py:214-249) for every env (or those with mask != 0).  Returns obs."""
        torch, dev = self._torch()
        if obs is None:
            obs = (torch.zeros if mask is not None else torch.empty)(
                (self.E, self.N, self.V, self.V, 3), dtype=torch.uint8, device=dev)
        if mask is not None:
            self._check_tensor(mask, (self.E,), torch.uint8, "mask")
        _capi.check(self._L.ssd_reset(self._h, self._dp(mask), self._dp(obs), self._obs_flags(obs), self._stream()), self._h)
        self.steps_since_full_reset = 0 if mask is None else None
        return obs

    def step(self, actions, order=None, out=None, auto_reset=False):
        """MapEnv.step (map_env.py:152-212) on every env.  actions: i32 [E,N] (-1 = absent);
        order: optional u8 [E,N] action-dict order.  Returns (obs, rew, done) device tensors.
        auto_reset: envs that reach the horizon (set_horizon) are reset by the same launch, their obs rows are the
        reset's (SSD_AUTO_RESET; uint8 observations only)."""
        torch, dev = self._torch()
        self._check_tensor(actions, (self.E, self.N), torch.int32, "actions")
        if order is not None:
            self._check_tensor(order, (self.E, self.N), torch.uint8, "order")
        # (back-to-back calls are bound by the host -- tools/step_rate.py -- so the output buffers of the last call again skip
        # their checks and pointer conversions, as in step_random())
        if out is not None and out is self._out_cache[0]:
            obs, rew, done = out
            po, pr, pd, fl = self._out_cache[1]
        else:
            obs, rew, done = out if out is not None else self.alloc_outputs()
            po, pr, pd, fl = self._dp(obs), self._dp(rew), self._dp(done), self._obs_flags(obs)
            self._out_cache = (out, (po, pr, pd, fl))
        rc = self._L.ssd_step(self._h, C.c_void_p(actions.data_ptr()), self._dp(order), po, pr, pd,
                              fl | (_capi.SSD_AUTO_RESET if auto_reset else 0), self._stream())
        if rc:
            _capi.check(rc, self._h)
        self._count_after_step(auto_reset)
        return obs, rew, done

    def _count_after_step(self, auto_reset):
        self._count_steps(1)
        if auto_reset and self.steps_since_full_reset is not None and getattr(self, "horizon", 0) > 0 \
                and self.steps_since_full_reset >= self.horizon:
            self.steps_since_full_reset = 0          # everybody reached the horizon together and was reset by that launch

    def step_random(self, out=None, actions_out=None, num_actions=None, auto_reset=False):
        """One step with uniform random actions drawn on the device (rollout.py:62-70)."""
        if out is not None and out is self._out_cache[0]:                  # same buffers as last time: pointers are known
            obs, rew, done = out
            po, pr, pd, fl = self._out_cache[1]
        else:
            obs, rew, done = out if out is not None else self.alloc_outputs()
            po, pr, pd, fl = self._dp(obs), self._dp(rew), self._dp(done), self._obs_flags(obs)
            self._out_cache = (out, (po, pr, pd, fl))
        na = self.num_actions if num_actions is None else int(num_actions)
        rc = self._L.ssd_step_random(self._h, na, self._dp(actions_out), po, pr, pd,
                                     fl | (_capi.SSD_AUTO_RESET if auto_reset else 0), self._stream())
        if rc:
            _capi.check(rc, self._h)
        self._count_after_step(auto_reset)
        return obs, rew, done

    def _count_steps(self, n):
        if self.steps_since_full_reset is not None:
            self.steps_since_full_reset += n

    def set_rollout_chains(self, chains):
        """How many independent env ranges rollout_random() enqueues on streams of its own (0 = automatic)."""
        _capi.check(self._L.ssd_set_rollout_chains(self._h, int(chains)), self._h)

    def _rollout_args(self, obs, rew, done):
        # (the buffers of the last call again: their checks and pointers are known -- a short call is mostly fixed costs)
        cache = getattr(self, "_roll_cache", None)
        if cache is not None and cache[0] is obs and cache[1] is rew and cache[2] is done:
            return cache[3]
        torch, dev = self._torch()
        first = obs if obs is not None else rew if rew is not None else done
        if first is None:
            raise ValueError("a rollout needs at least one of obs / rew / done (their leading dimension is the output ring)")
        ring = int(first.shape[0])
        if obs is not None:
            if obs.dtype not in (torch.uint8, torch.float32):
                raise ValueError("obs must be uint8 or float32")
            self._check_tensor(obs, (ring, self.E, self.N, self.V, self.V, 3), obs.dtype, "obs")
        if rew is not None:
            self._check_tensor(rew, (ring, self.E, self.N), torch.int32, "rew")
        if done is not None:
            self._check_tensor(done, (ring, self.E, self.N), torch.uint8, "done")
        args = (self._dp(obs), self._dp(rew), self._dp(done), ring, (_capi.SSD_OBS_F32 if obs is not None and obs.dtype == torch.float32 else 0))
        self._roll_cache = (obs, rew, done, args)
        return args

    def _count_rollout(self, n_steps, reset_every, step0):
        n_steps, reset_every, step0 = int(n_steps), int(reset_every), int(step0)
        last = None                                  # index of the last step of this call that a full reset preceded
        if reset_every > 0 and n_steps > 0:
            k = (n_steps - 1) - ((step0 + n_steps - 1) % reset_every)
            last = k if k >= 0 else None
        if last is not None:
            self.steps_since_full_reset = n_steps - last
        else:
            self._count_steps(n_steps)

    def rollout_random(self, n_steps, obs, rew=None, done=None, reset_every=0, step0=0, num_actions=None, fused=False):
        """rollout.py:58-70 as ONE library call: `n_steps` random-action steps (plus a full reset whenever
        (step0 + k) % reset_every == 0) enqueued back to back.  obs / rew / done are device tensors with a leading ring
        dimension R: step k writes slot (step0 + k) % R  (obs u8 or f32 [R,E,N,V,V,3], rew i32 [R,E,N], done u8 [R,E,N]).
        Same launches as n_steps calls of step_random(); the host just stops being the bottleneck.
        fused=True: ONE kernel launch for the whole call (SSD_ROLLOUT_FUSED) -- every env stays in LDS / registers across
        its steps; same results, uint8 observations only.  fused="auto": the library picks (SSD_ROLLOUT_AUTO: the fused kernel
        for uint8 observations and two steps or more, the chains otherwise; rollout_path() says which form ran)."""
        po, pr, pd, ring, f32 = self._rollout_args(obs, rew, done)
        na = self.num_actions if num_actions is None else int(num_actions)
        rc = self._L.ssd_rollout_random(self._h, na, int(n_steps), int(reset_every), int(step0), po, pr, pd, ring,
                                        f32 | self._fused_flag(fused), self._stream())
        if rc:
            _capi.check(rc, self._h)
        self._count_rollout(n_steps, reset_every, step0)

    def rollout_actions(self, actions, n_steps, obs, rew=None, done=None, reset_every=0, step0=0, fused=False, order=None):
        """The same call with caller-supplied actions (ssd_rollout_actions): what the reference's callers do, env.step(policy
        actions) per step (visuallizer_rllib.py:121-153), for a recorded sequence / an action chunk of n_steps steps.
        actions: int32 [A,E,N] device tensor (-1 = the agent does not act); step k reads slot (step0 + k) % A.  order: optional
        uint8 [A,E,N], per step the agent indices in action-dict order, 0xFF-terminated (None: index order).  Outputs as
        rollout_random() (fused=True / "auto" as there).  Reuse the same action / output tensors from call to call: the launches'
        arguments are cached by them."""
        torch, dev = self._torch()
        cache = getattr(self, "_act_cache", None)
        if cache is not None and cache[0] is actions and cache[1] is order:
            pa, pord, aring = cache[2]
        else:
            aring = int(actions.shape[0])
            self._check_tensor(actions, (aring, self.E, self.N), torch.int32, "actions")
            if order is not None:
                self._check_tensor(order, (aring, self.E, self.N), torch.uint8, "order")
            pa, pord = self._dp(actions), self._dp(order)
            self._act_cache = (actions, order, (pa, pord, aring))
        po, pr, pd, ring, f32 = self._rollout_args(obs, rew, done)
        rc = self._L.ssd_rollout_actions(self._h, pa, pord, aring, int(n_steps), int(reset_every), int(step0), po, pr, pd, ring,
                                         f32 | self._fused_flag(fused), self._stream())
        if rc:
            _capi.check(rc, self._h)
        self._count_rollout(n_steps, reset_every, step0)

    @staticmethod
    def _fused_flag(fused):
        if isinstance(fused, str):
            if fused != "auto":
                raise ValueError("fused must be True, False or 'auto'")
            return _capi.SSD_ROLLOUT_AUTO
        return _capi.SSD_ROLLOUT_FUSED if fused else 0

    def rollout_path(self):
        """How the last rollout call was dispatched (ssd_rollout_path): {"aql", "coherent", "split", "fused", "sync", "forked",
        "queue_dropped": bool, "chains": n, "pool": dispatch queues the device's pool settled on, "agent_match": how the HSA agent
        of the handle's HIP device was found ("pci" / "uuid" / "ordinal" / "none")} -- lets a benchmark or a test
        tell a silent fallback from the path it meant to measure."""
        m = self._L.ssd_rollout_path(self._h)
        return {"aql": bool(m & _capi.SSD_PATH_AQL), "coherent": bool(m & _capi.SSD_PATH_COHERENT), "split": bool(m & _capi.SSD_PATH_SPLIT),
                "fused": bool(m & _capi.SSD_PATH_FUSED), "sync": bool(m & _capi.SSD_PATH_SYNC), "forked": bool(m & _capi.SSD_PATH_FORKED),
                "queue_dropped": bool(m & _capi.SSD_PATH_QUEUE_DROPPED), "chains": (m >> 8) & 15, "pool": (m >> 12) & 7,
                "agent_match": ("none", "pci", "uuid", "ordinal")[(m >> 16) & 3]}

    def observe(self, rotate=True, obs=None):
        torch, dev = self._torch()
        if obs is None:
            obs = torch.empty((self.E, self.N, self.V, self.V, 3), dtype=torch.uint8, device=dev)
        _capi.check(self._L.ssd_observe(self._h, self._dp(obs), (0 if rotate else _capi.SSD_NO_ROTATE) | self._obs_flags(obs),
                                        self._stream()), self._h)
        return obs

    # ------------------------------------------------------------------ host-array API
    def _host_out(self):
        return (np.zeros((self.E, self.N, self.V, self.V, 3), np.uint8), np.zeros((self.E, self.N), np.int32),
                np.zeros((self.E, self.N), np.uint8))

    def reset_host(self, mask=None):
        obs = np.zeros((self.E, self.N, self.V, self.V, 3), np.uint8)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.E)
        _capi.check(self._L.ssd_reset(self._h, _ptr(m), _ptr(obs), _capi.SSD_HOST_PTRS, None), self._h)
        self.steps_since_full_reset = 0 if mask is None else None
        return obs

    def step_host(self, actions, order=None):
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint8).reshape(self.E, self.N)
        obs, rew, done = self._host_out()
        _capi.check(self._L.ssd_step(self._h, _ptr(actions), _ptr(order), _ptr(obs), _ptr(rew), _ptr(done),
                                     _capi.SSD_HOST_PTRS, None), self._h)
        self._count_steps(1)
        return obs, rew, done

    def step_random_host(self):
        act = np.zeros((self.E, self.N), np.int32)
        obs, rew, done = self._host_out()
        _capi.check(self._L.ssd_step_random(self._h, self.num_actions, _ptr(act), _ptr(obs), _ptr(rew), _ptr(done),
                                            _capi.SSD_HOST_PTRS, None), self._h)
        self._count_steps(1)
        return act, obs, rew, done

    def observe_host(self, rotate=True):
        obs = np.zeros((self.E, self.N, self.V, self.V, 3), np.uint8)
        flags = _capi.SSD_HOST_PTRS | (0 if rotate else _capi.SSD_NO_ROTATE)
        _capi.check(self._L.ssd_observe(self._h, _ptr(obs), flags, None), self._h)
        return obs

    # ------------------------------------------------------------------ state access
    def get_state(self):
        E, N, H, W = self.E, self.N, self.H, self.W
        s = dict(world=np.zeros((E, H, W), np.int8), beam=np.zeros((E, H, W), np.int8) if self.keep_beams else None,
                 pos=np.zeros((E, N, 2), np.int16), orient=np.zeros((E, N), np.uint8),
                 episode=np.zeros(E, np.uint32), t=np.zeros(E, np.uint32))
        _capi.check(self._L.ssd_get_state(self._h, _ptr(s["world"]), _ptr(s["beam"]), _ptr(s["pos"]), _ptr(s["orient"]),
                                          _ptr(s["episode"]), _ptr(s["t"])), self._h)
        return s

    def set_state(self, world=None, beam=None, pos=None, orient=None, episode=None, t=None):
        E, N, H, W = self.E, self.N, self.H, self.W

        def prep(a, dt, shape, name):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            if a.shape != shape:
                raise ValueError("%s must have shape %s, got %s" % (name, shape, a.shape))
            return a
        world, beam = prep(world, np.int8, (E, H, W), "world"), prep(beam, np.int8, (E, H, W), "beam")
        pos, orient = prep(pos, np.int16, (E, N, 2), "pos"), prep(orient, np.uint8, (E, N), "orient")
        episode, t = prep(episode, np.uint32, (E,), "episode"), prep(t, np.uint32, (E,), "t")
        _capi.check(self._L.ssd_set_state(self._h, _ptr(world), _ptr(beam), _ptr(pos), _ptr(orient), _ptr(episode),
                                          _ptr(t)), self._h)
        if t is not None:
            self.steps_since_full_reset = None

    def set_horizon(self, horizon):
        """done = (t >= horizon) from now on (RLlib's `horizon`, train_baseline.py:131); 0 = never (reference envs)."""
        _capi.check(self._L.ssd_set_horizon(self._h, int(horizon)), self._h)
        self.horizon = int(horizon)

    def waste_count(self):
        """u32 [E]: the number of 'H' cells the last step / reset computed the Cleanup spawn probabilities from."""
        out = np.zeros(self.E, np.uint32)
        _capi.check(self._L.ssd_get_waste_count(self._h, _ptr(out)), self._h)
        return out

    def render_full(self, e=0):
        """map_to_colors() of the whole grid of env e (map_env.py:316-339): u8 [H,W,3]."""
        rgb = np.zeros((self.H, self.W, 3), np.uint8)
        _capi.check(self._L.ssd_render_full(self._h, int(e), _ptr(rgb)), self._h)
        return rgb

    def render_frames(self, e_begin=0, count=None, out=None, host=False):
        """map_to_colors() of the whole grids of envs [e_begin, e_begin+count) in one launch: u8 [count,H,W,3] -- a device
        tensor (enqueued on the current stream), or with host=True a NumPy array (synchronous).  These are the frames
        rollout.py:77 / visuallizer_rllib.py:161 collect one env at a time."""
        count = self.E - e_begin if count is None else int(count)
        if e_begin < 0 or count < 0 or e_begin + count > self.E:
            raise ValueError("env range [%d, %d) outside the batch of %d" % (e_begin, e_begin + count, self.E))
        shape = (count, self.H, self.W, 3)
        if host:
            rgb = np.zeros(shape, np.uint8) if out is None else out
            if rgb.dtype != np.uint8 or rgb.shape != shape or not rgb.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous uint8 array of shape %s" % (shape,))
            if count:
                _capi.check(self._L.ssd_render_frames(self._h, int(e_begin), count, _ptr(rgb), _capi.SSD_HOST_PTRS, None), self._h)
            return rgb
        torch, dev = self._torch()
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=dev)
        elif out.dtype != torch.uint8 or tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor of shape %s" % (shape,))
        if count:
            _capi.check(self._L.ssd_render_frames(self._h, int(e_begin), count, self._dp(out), 0, self._stream()), self._h)
        return out

    def agent_action_obs(self, actions=None, done=None, out=None):
        """The `other_agent_actions` / `visible_agents` members of the reference's observation dict under
        return_agent_actions=True (map_env.py:201-205, :242-246, :749-770) for the whole batch: int64 [E,N,N-1] device tensors.
        actions: this step's int32 [E,N] actions (None: the reset form, zeros); done: optional u8 [E,N] (rows of envs that the
        step reset, or is about to, are zeros).  Row (e,i) lists the OTHER agents' actions in string-sorted id order, -1 for an
        agent that did not act.  visible_agents is all ones (the reference's quirk, :767)."""
        torch, dev = self._torch()
        shape = (self.E, self.N, max(self.N - 1, 0))
        if actions is not None:
            self._check_tensor(actions, (self.E, self.N), torch.int32, "actions")
        if done is not None:
            self._check_tensor(done, (self.E, self.N), torch.uint8, "done")
        oaa, vis = out if out is not None else (torch.empty(shape, dtype=torch.int64, device=dev), torch.empty(shape, dtype=torch.int64, device=dev))
        for t, name in ((oaa, "other_agent_actions"), (vis, "visible_agents")):
            self._check_tensor(t, shape, torch.int64, name)
        _capi.check(self._L.ssd_agent_action_obs(self._h, self._dp(actions), self._dp(done), self._dp(oaa), self._dp(vis), 0,
                                                 self._stream()), self._h)
        return oaa, vis

    def agent_action_obs_host(self, actions=None, done=None):
        shape = (self.E, self.N, max(self.N - 1, 0))
        oaa, vis = np.zeros(shape, np.int64), np.zeros(shape, np.int64)
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N)
        d = None if done is None else np.ascontiguousarray(done, dtype=np.uint8).reshape(self.E, self.N)
        _capi.check(self._L.ssd_agent_action_obs(self._h, _ptr(a), _ptr(d), _ptr(oaa), _ptr(vis), _capi.SSD_HOST_PTRS, None), self._h)
        return oaa, vis

    def status(self, clear=True):
        st = C.c_uint32(0)
        _capi.check(self._L.ssd_device_status(self._h, C.byref(st), int(clear)), self._h)
        return st.value

    def synchronize(self):
        _capi.check(self._L.ssd_synchronize(self._h), self._h)

    # ------------------------------------------------------------------ bookkeeping for bench / roofline
    def algorithmic_bytes_per_env_step(self):
        return cfgmod.algorithmic_bytes_per_env_step(self.H, self.W, self.N, self.V)
