# environment.py -- MethaneEnv with the reference's interface (PPOV2.0/environment.py:18-169),
# stepped by the HIP environment kernels (csrc/env.hip) as a 1-env batch.
#
# Signatures kept: MethaneEnv(), reset() -> float32[6], step(int) -> (obs, reward, done, info);
# attributes agent_pos, source_pos, conc_field, tke_field, trajectory, current_radius,
# explore_bonus, step_count, visited-equivalent, gaussian_params (PPOV2.1), action_space.n.
# `VecMethaneEnv` (uavppo/vec_env.py) is the same environment for N envs per launch.
import numpy as np
import torch

from config import (CONC_REWARD_COEF, DEVICE, ENV_VARIANT, EXPLORE_BONUS, GAUSSIAN_RADIUS, GRID_DIVISIONS, GRID_SIZE,
                    INITIAL_RADIUS, MAX_STEPS, MIN_RADIUS, PEAK_CONCENTRATION, RADIUS_DECAY, SEED,
                    TKE_PENALTY_FACTOR)  # noqa: F401
from uavppo import ops
from uavppo.vec_env import INFO_KEYS, VecMethaneEnv


class _Discrete:                 # stand-in for gym.spaces.Discrete (only `.n` is ever read)
    def __init__(self, n):
        self.n = n


class _Box:                      # stand-in for gym.spaces.Box
    def __init__(self, low, high, dtype):
        self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape


class MethaneEnv:
    def __init__(self, variant=None, device=None, seed=None, bank=None, bank_sources=None):
        self.source_pos = None
        self.grid_size = GRID_SIZE
        self.action_space = _Discrete(5)
        self.observation_space = _Box(np.zeros(6, np.float32), np.ones(6, np.float32), np.float32)
        self.current_radius = INITIAL_RADIUS
        self.min_radius = MIN_RADIUS
        self.radius_decay = RADIUS_DECAY
        self.cell_size = self.grid_size // GRID_DIVISIONS
        self.explore_bonus = EXPLORE_BONUS
        if seed is None:             # follow numpy's global RNG like the reference does, so np.random.seed() pins a run
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self._vec = VecMethaneEnv(1, variant or ENV_VARIANT, device or DEVICE, seed=seed, bank=bank,
                                  bank_sources=bank_sources)
        self._act = torch.zeros(1, dtype=torch.int32, device=self._vec.device)
        self._field = None
        self._started = False
        self.reset()

    # -- reference attributes computed on demand ------------------------------------------------
    def _sync_scalars(self):
        pos, src, steps, _ = self._vec.peek()
        self.agent_pos = pos[0].cpu().numpy()
        self.source_pos = src[0].cpu().numpy()
        self.step_count = int(steps[0])

    def _fields(self):
        if self._field is None:
            self._field = ops.env_materialise(self._vec.state, 1, self._vec.cfg(), 0).cpu().numpy()
        return self._field

    @property
    def conc_field(self):
        return self._fields()[..., 0]

    @property
    def tke_field(self):
        return self._fields()[..., 1]

    @property
    def gaussian_params(self):       # PPOV2.1/environment.py:64-69
        return {"mu_x": self.source_pos[0], "mu_y": self.source_pos[1], "sigma": GAUSSIAN_RADIUS,
                "peak": PEAK_CONCENTRATION}

    # -- gym-style API ---------------------------------------------------------------------------
    def reset(self):
        v = self._vec
        v.current_radius, v.explore_bonus = self.current_radius, self.explore_bonus
        if not self._started:
            obs = v.reset()
            self._started = True
        else:
            # the kernel already started the next episode when the last step ended one (auto-reset);
            # an explicit reset in the middle of an episode advances to a fresh episode
            if not getattr(self, "_just_ended", False):
                self._force_new_episode()
            obs = v.obs
        self._just_ended = False
        self._field = None
        self.trajectory = []
        self._sync_scalars()
        return obs[0].cpu().numpy()

    def _force_new_episode(self):
        """Abandon the running episode: step with radius = +inf so the kernel's auto-reset fires."""
        v = self._vec
        keep = v.current_radius
        v.current_radius = 1e30
        v.step(self._act)
        v.current_radius = keep

    def step(self, action, noise=None):
        """noise: optional two standard normals replacing the draw of environment.py:101 (parity tests)."""
        v = self._vec
        v.current_radius, v.explore_bonus = self.current_radius, self.explore_bonus
        self._act.fill_(int(action))
        v.step(self._act, None if noise is None else torch.as_tensor(noise, dtype=torch.float64).reshape(1, 2).to(v.device))
        obs = v.term_obs[0].cpu().numpy()
        reward = float(v.rew64[0])
        flags = int(v.flags[0])
        done, reached = bool(flags & 1), bool(flags & 2)
        info_t = v.info[0].cpu().numpy()
        if done:
            # state of the ENDED episode for the caller (train_ppo2.0.py:169,201-205), fields included
            self.agent_pos = np.array([obs[0] * GRID_SIZE, obs[1] * GRID_SIZE], dtype=np.float32)
            self.step_count += 1
            self._just_ended = True
        else:
            self._sync_scalars()
        self.trajectory.append({"pos": self.agent_pos.copy(), "conc": obs[2], "tke": obs[3], "reached": reached})
        info = {k: float(info_t[i]) for i, k in enumerate(INFO_KEYS)}
        return obs, reward, done, info
