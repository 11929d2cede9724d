"""Batched adapter for rollouts: E envs stepped by one kernel launch, per-env
episode horizon with automatic reset, observations left on the GPU.

This is what removes the Python-dict bottleneck of running E `MapEnv` objects: RLlib's `horizon: 1000`
lives in the launcher, not in the env (run_scripts/train_baseline.py:131, train_moa.py:122), so here the
kernel reports `done = (t >= horizon)` and the adapter resets exactly those envs with a masked reset --
the returned observation row of a finished env is the first observation of its next episode (the usual
vector-env auto-reset convention).

Two surfaces:
  * tensors: `reset()` / `step(actions)` -> torch tensors on the engine's device;
  * RLlib `BaseEnv`-style `poll()` / `send_actions()` / `try_reset()` with {env_id: {agent_id: ...}} dicts,
    for code written against that interface (a host round trip per call: use it for compatibility,
    not for speed).
"""
import numpy as np

from . import constants as K
from .engine import VecEngine

_NORMALISE = (np.arange(256, dtype=np.float64) - 128.0) / 255.0


class SSDVectorEnv(object):
    def __init__(self, game, num_envs, num_agents, horizon=1000, ascii_map=None, seed=0, device=0,
                 env_index_base=0, view_len=K.VIEW_LEN, float32_obs=False):
        self.engine = VecEngine(game, ascii_map, num_envs=num_envs, num_agents=num_agents, seed=seed, device=device,
                                env_index_base=env_index_base, view_len=view_len)
        self.num_envs, self.num_agents, self.horizon = num_envs, num_agents, int(horizon)
        self.float32_obs = bool(float32_obs)     # the kernel writes float32((u8 - 128) / 255) NHWC directly
        self.engine.set_horizon(self.horizon)
        self.agent_ids = ['agent-%d' % i for i in range(num_agents)]
        self._out = None
        self._pending = None

    # ------------------------------------------------------------------ tensor API
    def reset(self):
        self._out = self.engine.alloc_outputs(float32=self.float32_obs)
        self.engine.reset(obs=self._out[0])
        return self._out[0]

    # Auto-reset, three ways.  (1) While every env was last reset by the same call (the engine keeps count) they all reach the
    # horizon on the same step and the host knows which one: nothing to launch until then, a full reset then.  (2) Otherwise
    # the step launch itself resets the envs that finish (SSD_AUTO_RESET).  (3) float32 observations, which that kernel does
    # not write: a masked reset launch after every step (a masked reset only touches envs whose flag is set, so no host
    # synchronisation is needed to decide whether anything finished).
    def _in_kernel(self):
        return (self.horizon > 0 and self.num_agents > 0 and self.engine.steps_since_full_reset is None
                and not self.float32_obs)

    def _auto_reset(self, obs, done, in_kernel):
        if self.horizon <= 0 or not self.num_agents or in_kernel:
            return
        since = self.engine.steps_since_full_reset
        if since is not None:
            if since >= self.horizon:
                self.engine.reset(obs=obs)           # everybody just finished: a full reset, no mask needed
            return
        self.engine.reset(mask=done[:, 0].contiguous(), obs=obs)

    def step(self, actions):
        """actions: int32 [E,N] on the device.  Returns (obs u8, rew i32, done u8) device tensors; envs whose
        episode just ended have been reset and their obs rows replaced by the new episode's first observation."""
        in_kernel = self._in_kernel()
        obs, rew, done = self.engine.step(actions, out=self._out, auto_reset=in_kernel)
        self._auto_reset(obs, done, in_kernel)
        return obs, rew, done

    def step_random(self):
        in_kernel = self._in_kernel()
        obs, rew, done = self.engine.step_random(out=self._out, auto_reset=in_kernel)
        self._auto_reset(obs, done, in_kernel)
        return obs, rew, done

    @staticmethod
    def to_float(obs):
        """uint8 [E,N,V,V,3] -> float32 NHWC batch [(E*N),V,V,3] with the reference's scaling
        ((x - 128) / 255, map_env.py:199), on the device: the input the first conv layer of
        models/conv_to_fcnet_v2.py:33-56 expects.  With float32_obs=True the step
        kernel writes this directly and no conversion pass is needed."""
        import torch
        E, N, V = obs.shape[0], obs.shape[1], obs.shape[2]
        return ((obs.to(torch.float32) - 128.0) / 255.0).reshape(E * N, V, V, 3)

    # ------------------------------------------------------------------ BaseEnv-style API
    def poll(self):
        """-> (obs, rewards, dones, infos, off_policy_actions) as {env_id: {agent_id: value}} dicts."""
        if self.float32_obs:
            raise RuntimeError("the dict surface renormalises uint8 observations: construct with float32_obs=False")
        if self._pending is None:
            obs = self.reset().cpu().numpy()
            rew = np.zeros((self.num_envs, self.num_agents), np.int32)
            done = np.zeros((self.num_envs, self.num_agents), np.uint8)
        else:
            obs, rew, done = (x.cpu().numpy() for x in self._pending)
        o, r, d, i = {}, {}, {}, {}
        for e in range(self.num_envs):
            o[e] = {a: _NORMALISE[obs[e, k]] for k, a in enumerate(self.agent_ids)}
            r[e] = {a: int(rew[e, k]) for k, a in enumerate(self.agent_ids)}
            d[e] = {a: bool(done[e, k]) for k, a in enumerate(self.agent_ids)}
            d[e]["__all__"] = bool(done[e].any()) if self.num_agents else False
            i[e] = {}
        return o, r, d, i, {}

    def send_actions(self, action_dict):
        import torch
        act = np.full((self.num_envs, self.num_agents), K.NO_ACTION, np.int32)
        for e, per_agent in action_dict.items():
            for a, v in per_agent.items():
                act[e, self.agent_ids.index(a)] = int(v)
        dev = torch.device("cuda", self.engine.device)
        self._pending = self.step(torch.from_numpy(act).to(dev))

    def try_reset(self, env_id):
        import torch
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=torch.device("cuda", self.engine.device))
        mask[env_id] = 1
        if self._out is None:
            self._out = self.engine.alloc_outputs(float32=self.float32_obs)
        self.engine.reset(mask=mask, obs=self._out[0])
        return {a: _NORMALISE[self._out[0][env_id, k].cpu().numpy()] for k, a in enumerate(self.agent_ids)}
