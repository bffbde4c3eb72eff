"""CPU oracle of the PROCEDURAL-field mode (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

The reference fills a fresh 500x500 table per reset from numpy's MT19937 (`rand(2)`, `randn(G,G)`, `rand(G,G)`,
PPOV2.0/environment.py:41-62) and draws `randn(2)` per step (:101).  The product's default mode keeps no table: the
draws of (env, episode, cell) / (env, episode, step) come from a counter RNG (Philox4x32-10, Salmon et al. SC'11) and
the cell value is regenerated on demand.  The random STREAM therefore cannot equal numpy's; everything ELSE can, and this
module restates it so that the mode `bench.py` measures is pinned cell for cell and step for step:

  * `philox4x32_10`       the generator, with the product's keying (seed = key, counter = (index, env, episode|iter, purpose));
                          pinned to the Random123 known-answer vectors (tests/test_oracle_procedural.py)
  * `source_of`           E2  `rand(2)*(500-100)+50`                  PPOV2.0/environment.py:42-43
  * `field_cells`         E3  `base + turbulence`, clipped             PPOV2.0/environment.py:51-62, PPOV2.1/environment.py:52-61
                          f64, the reference's own operation order (sqrt then square, (|g| + ripple) + 0.2 u), the SAME
                          expression as `env_oracle.make_fields`, which is bit-exact against the reference's tables
  * `step_normals`        E4  the `randn(2)` of environment.py:101
  * `action_uniform`      R1  the uniform behind `Categorical.sample()`  train_ppo2.0.py:162

The normals are Box-Muller on 24-bit uniforms, z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = ... sin(2 pi u2) with u1 in (0,1],
u2 in [0,1) -- the definition the device code uses (csrc/env_core.h `normal2`, `field_at`), here in numpy f64.
`ProceduralVecEnv` = N auto-resetting `EnvCore`s (the reference-pinned E4/E5 arithmetic) over lazily evaluated fields.
"""
from __future__ import annotations

import numpy as np

from .env_oracle import CONC_PEAK, EXPLORE_BONUS, GRID, INITIAL_RADIUS, PADDING, TURB_INT, VARIANTS, EnvCore

RNG_SOURCE, RNG_FIELD, RNG_STEP, RNG_ACTION = 1, 2, 3, 4     # csrc/philox.h
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(seed, c0, c1, c2, c3):
    """Philox4x32-10.  seed: python int (64-bit key = (lo, hi)); c0..c3: ints or integer arrays (broadcast).
    Returns four uint64 arrays holding 32-bit words."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) & _LO for c in (c0, c1, c2, c3)])
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0                       # < 2^64: both factors are 32-bit
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _LO
        hi1, lo1 = p1 >> _S32, p1 & _LO
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u53(a, b):
    """53-bit uniform in [0,1) from two 32-bit words (csrc/philox.h u01_f64)."""
    return (((a << _S32) | b) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _bm_radius(a):
    u1 = ((a >> np.uint64(8)).astype(np.float64) + 1.0) * (1.0 / 16777216.0)          # (0,1]
    return np.sqrt(-2.0 * np.log(u1))


def _u24(a):
    return (a >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)                # [0,1)


def source_of(seed, env, episode):
    """E2: source position of episode `episode` of global env `env` (environment.py:42-43)."""
    x, y, z, w = philox4x32_10(seed, 0, env, episode, RNG_SOURCE)
    pad = PADDING
    return np.stack([_u53(x, y) * (GRID - 2 * pad) + pad, _u53(z, w) * (GRID - 2 * pad) + pad], axis=-1)


def field_cells(seed, env, episode, source, sigma, x, y):
    """E3 at integer cells (x, y) (arrays): (conc, tke) in f64, environment.py:52-61 term for term."""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    r0, r1, _, r3 = philox4x32_10(seed, x * GRID + y, env, episode, RNG_FIELD)
    gauss = _bm_radius(r0) * np.cos(2.0 * np.pi * _u24(r1))              # randn(G,G)[x,y]
    unif = r3.astype(np.float64) * (1.0 / 4294967296.0)                  # rand(G,G)[x,y]
    dist = np.sqrt((x - source[0]) ** 2 + (y - source[1]) ** 2)          # :53
    base = CONC_PEAK * np.exp(-dist ** 2 / (2 * sigma ** 2))             # :54 / V2.1 :56
    turb = TURB_INT * (np.abs(gauss) + 0.3 * np.sin(0.05 * x) * np.cos(0.07 * y) + 0.2 * unif)   # :56-60
    return np.clip(base + turb, 0, CONC_PEAK), turb                      # :61-62


def full_field(seed, env, episode, sigma):
    """The whole table of one (env, episode), as the reference would hold it in conc_field / tke_field."""
    src = source_of(seed, env, episode)
    x, y = np.mgrid[:GRID, :GRID]
    conc, tke = field_cells(seed, env, episode, src, sigma, x, y)
    return src, conc, tke


def step_normals(seed, env, episode, step):
    """E4: the two normals of the step taken when the episode's step counter reads `step` (0 for the first)."""
    r0, r1, _, _ = philox4x32_10(seed, step, env, episode, RNG_STEP)
    rad, ang = _bm_radius(r0), 2.0 * np.pi * _u24(r1)
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)


def action_uniform(seed, t, env, iteration):
    """R1: the 24-bit uniform in [0,1) (f32) the rollout's Categorical draw of (step t, env, iteration) uses."""
    r0 = philox4x32_10(seed, t, env, iteration, RNG_ACTION)[0]
    return (r0 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def sample_inverse_cdf(probs, u):
    """torch.multinomial-style draw from unnormalised f32 probabilities: first a with u * sum < cumsum_a (f32)."""
    probs = np.asarray(probs, np.float32)
    psum = np.float32(0)
    for a in range(probs.shape[-1]):
        psum = psum + probs[..., a]
    target = np.asarray(u, np.float32) * psum
    cdf = np.zeros_like(target)
    sel = np.full(target.shape, probs.shape[-1] - 1, np.int64)
    found = np.zeros(target.shape, bool)
    for a in range(probs.shape[-1]):
        cdf = cdf + probs[..., a]
        hit = (~found) & (target < cdf)
        sel[hit] = a
        found |= hit
    return sel


class LazyField:
    """conc / tke 'tables' of one (env, episode) indexed [x, y], evaluated per cell on demand (memoised)."""

    def __init__(self, seed, env, episode, source, sigma, which):
        self.args = (seed, env, episode, source, sigma)
        self.which = which
        self.memo = {}

    def __getitem__(self, xy):
        v = self.memo.get(xy)
        if v is None:
            v = field_cells(*self.args, xy[0], xy[1])[self.which][()]
            self.memo[xy] = v
        return v


class ProceduralVecEnv:
    """N auto-resetting environments in procedural-field mode: per-env semantics are EnvCore's (reference-pinned),
    draws are the counter RNG's.  `env_offset` = global index of env 0 (multi-GPU sharding keys by global index)."""

    def __init__(self, n, seed, variant="v2.0", radius=INITIAL_RADIUS, bonus=EXPLORE_BONUS, trend_k=0, env_offset=0):
        self.n, self.seed, self.off = int(n), int(seed), int(env_offset)
        self.sigma = VARIANTS[variant][0]
        self.envs = [EnvCore(variant, trend_k) for _ in range(self.n)]
        self.obs_dim = 6 + trend_k
        self.episode = np.zeros(self.n, np.int64)
        self.set_curriculum(radius, bonus)

    def set_curriculum(self, radius, bonus):
        for e in self.envs:
            e.radius, e.bonus = radius, bonus

    def _begin(self, i):
        g, k = self.off + i, int(self.episode[i])
        src = source_of(self.seed, g, k)
        return self.envs[i].begin_episode(src, LazyField(self.seed, g, k, src, self.sigma, 0),
                                          LazyField(self.seed, g, k, src, self.sigma, 1))

    def reset(self):
        self.episode[:] = 0
        return np.stack([self._begin(i) for i in range(self.n)])

    def step(self, actions, normals=None):
        """normals=None: the counter RNG's own step noise (what the product does when none is injected)."""
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        term = np.zeros((self.n, self.obs_dim), np.float32)
        rew = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, bool)
        reached = np.zeros(self.n, bool)
        info = np.zeros((self.n, 5), np.float64)
        for i, e in enumerate(self.envs):
            z = normals[i] if normals is not None else step_normals(self.seed, self.off + i, int(self.episode[i]), e.steps)
            o, r, d, s, inf = e.step(int(actions[i]), z)
            term[i], rew[i], done[i], reached[i], info[i] = o, r, d, s, inf
            if d:
                self.episode[i] += 1
                o = self._begin(i)
            obs[i] = o
        return obs, rew, done, reached, info, term
