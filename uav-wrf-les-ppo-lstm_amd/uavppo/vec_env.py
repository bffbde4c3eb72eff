"""VecMethaneEnv: N independent MethaneEnv instances stepped by one HIP kernel launch.

Per-env semantics are the reference's (PPOV2.0/environment.py:18-169; sigma / clip / MAX_STEPS
variants of PPOV2.1 and PPOV1.1); vectorisation and auto-reset (the `env.reset()` of
train_ppo2.0.py:139 folded into the step) are the only additions.  Fields are either
procedural (counter RNG keyed by (env, episode, cell): O(1) memory) or a bank of materialised
[F,500,500,2] f64 tables resident in HBM (episode k of env e uses field (e + k*N) mod F).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

INFO_KEYS = ("concentration_reward", "explore_reward", "move_penalty", "tke_penalty", "boundary_penalty")


class VecMethaneEnv:
    def __init__(self, num_envs, variant="v2.0", device="cuda", seed=1234, bank=None, bank_sources=None,
                 env_offset=0, n_env_total=None, trend_k=0):
        self.num_envs = int(num_envs)
        self.variant = variant
        self.device = torch.device(device)
        self.seed = int(seed)
        self.env_offset = int(env_offset)
        self.n_env_total = int(n_env_total or num_envs)
        self.max_steps = ops.ENV_MAX_STEPS[variant]
        self.trend_k = int(trend_k)          # extra obs channels: obs[2](t) - obs[2](t-1-i)  (BASELINE C5)
        self.obs_dim = 6 + self.trend_k
        self.current_radius = 50.0           # INITIAL_RADIUS, config.py:27 / environment.py:31
        self.explore_bonus = 0.6             # EXPLORE_BONUS,  config.py:21 / environment.py:38
        self.bank = None if bank is None else torch.as_tensor(bank, dtype=torch.float64).to(self.device).contiguous()
        self.bank_sources = (None if bank_sources is None else
                             torch.as_tensor(bank_sources, dtype=torch.float64).to(self.device).contiguous())
        n, d = self.num_envs, self.device
        self.state = torch.zeros(ops.env_state_bytes(n), dtype=torch.uint8, device=d)
        self.obs = torch.zeros(n, self.obs_dim, dtype=torch.float32, device=d)
        self.rew = torch.zeros(n, dtype=torch.float32, device=d)
        self.rew64 = torch.zeros(n, dtype=torch.float64, device=d)
        self.done = torch.zeros(n, dtype=torch.float32, device=d)
        self.flags = torch.zeros(n, dtype=torch.uint8, device=d)
        self.info = torch.zeros(n, 5, dtype=torch.float32, device=d)
        self.term_obs = torch.zeros(n, self.obs_dim, dtype=torch.float32, device=d)

    def cfg(self):
        return ops.make_env_cfg(self.variant, self.current_radius, self.explore_bonus, self.seed, self.bank,
                                self.bank_sources, self.env_offset, self.n_env_total, self.trend_k)

    def reset(self):
        ops.env_reset(self.state, self.num_envs, self.cfg(), self.obs)
        return self.obs

    def step(self, actions, noise=None):
        """actions: int32 [N] on the device; noise: optional f64 [N,2] standard normals (parity tests).
        Returns (obs, reward, done, info) as device tensors; `obs` is the state to act on next
        (the reset observation where done), `self.term_obs` the observation of the ended step."""
        ops.env_step(self.state, self.num_envs, self.cfg(), actions, self.obs, self.rew, self.done, self.flags,
                     noise=noise, info=self.info, term_obs=self.term_obs, rew64=self.rew64)
        return self.obs, self.rew, self.done, self.info

    def peek(self):
        n, d = self.num_envs, self.device
        pos = torch.empty(n, 2, dtype=torch.float32, device=d)
        src = torch.empty(n, 2, dtype=torch.float64, device=d)
        steps = torch.empty(n, dtype=torch.int32, device=d)
        epi = torch.empty(n, dtype=torch.int32, device=d)
        ops.env_peek(self.state, n, pos, src, steps, epi)
        return pos, src, steps, epi
