"""Per-episode training log of the vectorised trainer: the 11 CSV columns of the reference
(PPOV2.0/train_ppo2.0.py:129-135, filled at :177-183, :200-206, :230-242), one row per finished episode in
(iteration, env, time) order.  Host-side numpy over the rollout buffers (cumulative sums along T; episodes may
span rollouts, so per-env partial sums are carried)."""
from __future__ import annotations

import numpy as np

COLUMNS = ["Episode", "Total_Reward", "Success", "Conc_Reward", "Explore_Reward", "Move_Penalty", "TKE_Penalty",
           "Boundary_Penalty", "Steps", "Final_Conc", "Current_Radius"]


class EpisodeLogger:
    def __init__(self, num_envs):
        self.carry = np.zeros((num_envs, 6), np.float64)     # running [total, conc, explore, move, tke, boundary]
        self.steps = np.zeros(num_envs, np.int64)
        self.count = 0
        self.rows = []

    def add_rollout(self, rew, info, flags, radius):
        """rew [N,T] f32, info [N,T,6] f32 (5 reward parts + obs[2] of the step), flags [N,T] u8."""
        rew, info, flags = np.asarray(rew), np.asarray(info), np.asarray(flags)
        N, T = rew.shape
        parts = np.concatenate([rew[..., None].astype(np.float64), info[..., :5].astype(np.float64)], axis=2)
        cs = np.cumsum(parts, axis=1)                                     # [N,T,6]
        n_idx, t_idx = np.nonzero(flags & 1)                              # ended steps, (env, time) order
        if n_idx.size:
            first = np.ones(n_idx.size, bool)
            first[1:] = n_idx[1:] != n_idx[:-1]
            prev_t = np.where(first, -1, np.roll(t_idx, 1))
            base = np.where(prev_t[:, None] >= 0, cs[n_idx, np.maximum(prev_t, 0)], 0.0)
            sums = cs[n_idx, t_idx] - base + np.where(first[:, None], self.carry[n_idx], 0.0)
            steps = (t_idx - prev_t) + np.where(first, self.steps[n_idx], 0)
            success = (flags[n_idx, t_idx] & 2) > 0
            final_conc = np.where(success, info[n_idx, t_idx, 5].astype(np.float64) * 100.0, 0.0)   # :203
            for k in range(n_idx.size):
                self.count += 1
                self.rows.append([self.count, sums[k, 0], int(success[k]), sums[k, 1], sums[k, 2], sums[k, 3], sums[k, 4],
                                  sums[k, 5], int(steps[k]), final_conc[k], radius])
        # carry what is left of each row after its last ended episode
        last = np.full(N, -1)
        if n_idx.size:
            last[n_idx] = t_idx                                           # later entries overwrite: last end per env
        tail = cs[:, -1] - np.where(last[:, None] >= 0, cs[np.arange(N), np.maximum(last, 0)], 0.0)
        self.carry = np.where(last[:, None] >= 0, tail, self.carry + tail)
        self.steps = np.where(last >= 0, T - 1 - last, self.steps + T)
