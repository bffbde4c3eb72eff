"""Vectorised CPU form of the plume-environment oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

`NumpyVecEnv` steps N environments with numpy array operations instead of a Python loop over `EnvCore`
objects; per-env semantics are EnvCore's, i.e. the reference's (oracle/env_oracle.py cites the lines:
E2 PPOV2.0/environment.py:41-49, E4 :82-169, E5 :64-80), with the same f32/f64 typing, so it is
bit-identical to `OracleVecEnv` (tests/test_oracle_vec_env.py).  It exists for ONE purpose: variant (ii) of the
CPU baseline of SURVEY 8(d) -- "vectorised CPU, same N and T as the GPU run, all cores" -- which bench.py times
next to the reference-faithful single-env loop.  Fields come from a `FieldBank` (materialised mode: episode k of
env e uses field (e + k*N) mod F, the product's rule), because filling a fresh 500x500 table per reset, as the
reference does, would cost 16 ms per episode and swamp everything else at thousands of environments.
"""
from __future__ import annotations

import numpy as np

from .env_oracle import (BOUNDARY_DECAY_START, BOUNDARY_PENALTY, CELL, CELLS, CONC_PEAK, CONC_REWARD_COEF,
                         EXPLORE_BONUS, GRID, INITIAL_RADIUS, MOVE, TKE_PENALTY_FACTOR, TURB_INT, VARIANTS, _DXY)


class NumpyVecEnv:
    def __init__(self, n, bank, variant="v2.0", radius=INITIAL_RADIUS, bonus=EXPLORE_BONUS, trend_k=0):
        self.n, self.bank = int(n), bank
        self.sigma, self.clip_hi, self.max_steps = VARIANTS[variant]
        self.trend_k = int(trend_k)
        self.obs_dim = 6 + self.trend_k
        self.radius, self.bonus = radius, bonus
        self.ar = np.arange(self.n)
        self.dxy = np.asarray(_DXY, np.float64)
        # float(vc) ** 0.75 of environment.py:133 through Python's own pow, tabulated over every possible count
        self.pow075 = np.array([float(v) ** 0.75 for v in range(max(self.max_steps, 1) + 2)], np.float64)
        self.pos = np.zeros((self.n, 2), np.float64)       # holds f32-rounded values after an env's first step
        self.steps = np.zeros(self.n, np.int64)
        self.episode = np.zeros(self.n, np.int64)
        self.field = np.zeros(self.n, np.int64)
        self.visited = np.zeros((self.n, CELLS, CELLS), np.int64)
        self.q = np.zeros((self.n, 2), np.float32)

    def set_curriculum(self, radius, bonus):
        self.radius, self.bonus = radius, bonus

    # ---- E2
    def _begin(self, idx):
        self.field[idx] = (idx + self.episode[idx] * self.n) % self.bank.F
        self.pos[idx] = 0.0
        self.steps[idx] = 0
        self.visited[idx] = 0
        o2 = (self.bank.conc[self.field[idx], 0, 0] / CONC_PEAK).astype(np.float32)
        self.q[idx, 0] = o2
        self.q[idx, 1] = o2

    def reset(self):
        self.episode[:] = 0
        self._begin(self.ar)
        return self.obs()

    @staticmethod
    def _cell(v):
        return np.clip(v.astype(np.int64), 0, GRID - 1)      # int(): truncation toward zero, then the clip

    # ---- E5
    def obs(self):
        p32 = self.pos.astype(np.float32)
        x, y = self._cell(p32[:, 0]), self._cell(p32[:, 1])
        vc = self.visited[self.ar, x // CELL, y // CELL]
        o = np.empty((self.n, self.obs_dim), np.float32)
        o[:, 0] = p32[:, 0] / GRID
        o[:, 1] = p32[:, 1] / GRID
        o[:, 2] = self.bank.conc[self.field, x, y] / CONC_PEAK
        o[:, 3] = self.bank.tke[self.field, x, y] / (TURB_INT * 3)
        o[:, 4] = self.steps / self.max_steps
        o[:, 5] = np.minimum(vc / 5.0, 1.0)
        for i in range(self.trend_k):
            o[:, 6 + i] = o[:, 2] - self.q[:, i]
        return o

    # ---- E4 + the auto-reset of train_ppo2.0.py:139
    def step(self, actions, normals):
        a = np.asarray(actions).astype(np.int64)
        z = np.asarray(normals, np.float64)
        self.steps += 1
        px, py = self._cell(self.pos[:, 0]), self._cell(self.pos[:, 1])
        conc_here = self.bank.conc[self.field, px, py]
        prev_conc = conc_here / CONC_PEAK
        d = self.dxy[a]
        norm_d = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        move_penalty = -0.15 * (1 - norm_d / MOVE)
        turb = MOVE * 0.2 * (z * self.bank.tke[self.field, px, py][:, None] / (TURB_INT * 3))
        new = np.clip(self.pos + d + turb, 0, self.clip_hi)
        o2_old = prev_conc.astype(np.float32)
        self.pos = new.astype(np.float32).astype(np.float64)
        self.q[:, 1] = self.q[:, 0]
        self.q[:, 0] = o2_old

        cx, cy = self._cell(new[:, 0]), self._cell(new[:, 1])
        cur_conc = self.bank.conc[self.field, cx, cy] / CONC_PEAK
        grad = (cur_conc - prev_conc) / (norm_d + 1e-6)
        bdist = np.minimum(np.minimum(new[:, 0] / GRID, (GRID - new[:, 0]) / GRID),
                           np.minimum(new[:, 1] / GRID, (GRID - new[:, 1]) / GRID))
        bpen = np.where((bdist < BOUNDARY_DECAY_START) & (grad < -0.01),
                        -BOUNDARY_PENALTY * (BOUNDARY_DECAY_START - bdist) ** 2, 0.0)
        gx, gy = (new[:, 0] // CELL).astype(np.int64), (new[:, 1] // CELL).astype(np.int64)
        self.visited[self.ar, gx, gy] += 1
        vc = self.visited[self.ar, gx, gy]

        o = self.obs()
        den = self.pow075[vc] + 1
        conc_r = np.float32(CONC_REWARD_COEF) * o[:, 2]
        tke_p = np.float32(TKE_PENALTY_FACTOR) * o[:, 3]
        if isinstance(self.bonus, np.floating) and not isinstance(self.bonus, np.float32):
            explore = (self.bonus * (np.float32(1) - o[:, 5]).astype(np.float64)) / den          # f64 (model.py:142)
            total = conc_r.astype(np.float64) + explore
        else:
            explore = (np.float32(self.bonus) * (np.float32(1) - o[:, 5])) / den.astype(np.float32)   # f32 throughout
            total = (conc_r + explore).astype(np.float64)
        total = total + move_penalty
        total = total - tke_p.astype(np.float64)
        total = total + bpen
        src = self.bank.sources[self.field]
        dd = self.pos - src
        dist = np.sqrt(dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1])
        reached = dist <= self.radius
        total = np.where(reached, total + min(500, 150 * (INITIAL_RADIUS / self.radius)), total)
        done = (self.steps >= self.max_steps) | reached
        info = np.stack([conc_r.astype(np.float64), np.asarray(explore, np.float64), move_penalty,
                         -tke_p.astype(np.float64), bpen], 1)
        term = o
        idx = np.nonzero(done)[0]
        if idx.size:
            self.episode[idx] += 1
            self._begin(idx)
            o = self.obs()
        return o, total, done, reached, info, term
