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
                 env_index_base=0, view_len=K.VIEW_LEN, float32_obs=False, return_agent_actions=False):
        self.engine = VecEngine(game, ascii_map, num_envs=num_envs, num_agents=num_agents, seed=seed, device=device,
                                env_index_base=env_index_base, view_len=view_len)
        self.num_envs, self.num_agents, self.horizon = num_envs, num_agents, int(horizon)
        self.float32_obs = bool(float32_obs)     # the kernel writes float32((u8 - 128) / 255) NHWC directly
        self.engine.set_horizon(self.horizon)
        self.agent_ids = ['agent-%d' % i for i in range(num_agents)]
        # return_agent_actions=True (the MOA trainers' envs, run_scripts/train_moa.py:70): observations become a dict
        # {"curr_obs", "other_agent_actions" int64 [E,N,N-1], "visible_agents" int64 [E,N,N-1]} of device tensors -- the
        # reference's per-agent observation dict (map_env.py:201-205, :242-246) batched (engine.agent_action_obs)
        self.return_agent_actions = bool(return_agent_actions)
        self._out = None
        self._extras = None
        self._act_buf = None
        self._pending = None

    # ------------------------------------------------------------------ tensor API
    def _wrap(self, obs, actions, done):
        if not self.return_agent_actions:
            return obs
        self._extras = self.engine.agent_action_obs(actions, done, out=self._extras)
        return {"curr_obs": obs, "other_agent_actions": self._extras[0], "visible_agents": self._extras[1]}

    def reset(self):
        self._out = self.engine.alloc_outputs(float32=self.float32_obs)
        self.engine.reset(obs=self._out[0])
        return self._wrap(self._out[0], None, None)

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

    def step(self, actions, order=None):
        """actions: int32 [E,N] on the device (-1: the agent sent no action); order: optional uint8 [E,N] on the device, per env
        the agent indices in action-dict order, 0xFF-terminated (None: index order).  Returns (obs u8, rew i32, done u8) device
        tensors; envs whose episode just ended have been reset and their obs rows replaced by the new episode's first observation."""
        in_kernel = self._in_kernel()
        obs, rew, done = self.engine.step(actions, order=order, out=self._out, auto_reset=in_kernel)
        self._auto_reset(obs, done, in_kernel)
        # (rows of envs whose episode just ended carry a reset's observation: their other_agent_actions are the reset's zeros)
        return self._wrap(obs, actions, done if self.horizon > 0 else None), rew, done

    def step_random(self):
        in_kernel = self._in_kernel()
        if self.return_agent_actions and self._act_buf is None:
            import torch
            self._act_buf = torch.empty((self.num_envs, self.num_agents), dtype=torch.int32, device=self._out[1].device)
        obs, rew, done = self.engine.step_random(out=self._out, actions_out=self._act_buf, auto_reset=in_kernel)
        self._auto_reset(obs, done, in_kernel)
        return self._wrap(obs, self._act_buf, done if self.horizon > 0 else None), rew, done

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
            obs = self.reset()
            rew = np.zeros((self.num_envs, self.num_agents), np.int32)
            done = np.zeros((self.num_envs, self.num_agents), np.uint8)
        else:
            obs, rew, done = self._pending
            rew, done = rew.cpu().numpy(), done.cpu().numpy()
        oaa = vis = None
        if self.return_agent_actions:
            oaa, vis = obs["other_agent_actions"].cpu().numpy(), obs["visible_agents"].cpu().numpy()
            obs = obs["curr_obs"]
        obs = obs.cpu().numpy()
        o, r, d, i = {}, {}, {}, {}
        for e in range(self.num_envs):
            if self.return_agent_actions:
                # (an agent that sent no action is absent from the reference's array: drop its -1)
                o[e] = {a: {"curr_obs": _NORMALISE[obs[e, k]], "other_agent_actions": oaa[e, k][oaa[e, k] >= 0],
                            "visible_agents": vis[e, k]} for k, a in enumerate(self.agent_ids)}
            else:
                o[e] = {a: _NORMALISE[obs[e, k]] for k, a in enumerate(self.agent_ids)}
            r[e] = {a: int(rew[e, k]) for k, a in enumerate(self.agent_ids)}
            d[e] = {a: bool(done[e, k]) for k, a in enumerate(self.agent_ids)}
            d[e]["__all__"] = bool(done[e].any()) if self.num_agents else False
            i[e] = {}
        return o, r, d, i, {}

    def send_actions(self, action_dict):
        """{env_id: {agent_id: action}}.  The reference's step depends on the iteration order of the inner dicts (who is
        shuffled where in update_moves, whose beam lands first: map_env.py:171,379,546), so the order of every env's dict goes to
        the kernel with the actions, as MapEnv.step passes its own (map_env.py of this package); envs whose dicts are in
        index order -- what RLlib's sampler sends -- need none, and a batch of only such envs takes the map-specific kernels."""
        import torch
        N = self.num_agents
        act = np.full((self.num_envs, N), K.NO_ACTION, np.int32)
        order = np.full((self.num_envs, N), 0xFF, np.uint8)
        index_of = {a: k for k, a in enumerate(self.agent_ids)}
        explicit = False
        for e, per_agent in action_dict.items():
            last = -1
            for k, (a, v) in enumerate(per_agent.items()):
                i = index_of[a]                                  # KeyError for an unknown agent, as in the reference
                act[e, i] = int(v)
                order[e, k] = i
                explicit = explicit or i < last
                last = i
        dev = torch.device("cuda", self.engine.device)
        if explicit:
            # (envs that sent nothing keep an empty order: nobody acts there, which is what their all -1 action rows say too)
            self._pending = self.step(torch.from_numpy(act).to(dev), torch.from_numpy(order).to(dev))
        else:
            self._pending = self.step(torch.from_numpy(act).to(dev))

    def try_reset(self, env_id):
        import torch
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=torch.device("cuda", self.engine.device))
        mask[env_id] = 1
        if self._out is None:
            self._out = self.engine.alloc_outputs(float32=self.float32_obs)
        self.engine.reset(mask=mask, obs=self._out[0])
        if self.return_agent_actions:
            n1 = max(self.num_agents - 1, 0)
            return {a: {"curr_obs": _NORMALISE[self._out[0][env_id, k].cpu().numpy()], "other_agent_actions": np.zeros(n1, np.int64),
                        "visible_agents": np.ones(n1, np.int64)} for k, a in enumerate(self.agent_ids)}
        return {a: _NORMALISE[self._out[0][env_id, k].cpu().numpy()] for k, a in enumerate(self.agent_ids)}
