"""The numpy-vectorised environment oracle (bench.py's vectorised CPU baseline) is bit-identical to the
per-env oracle loop, which tests/test_oracle_env.py pins to the reference's own traces."""
import numpy as np
import pytest

from oracle.env_oracle import FieldBank, OracleVecEnv
from oracle.vec_env_oracle import NumpyVecEnv


@pytest.mark.parametrize("variant,trend_k,bonus,radius", [("v2.0", 0, 0.6, 50.0), ("v2.1", 2, np.float64(0.43), 31.0),
                                                          ("v1.1", 1, 0.6, 60.0)])
def test_numpy_vec_env_equals_per_env_oracle(variant, trend_k, bonus, radius):
    N, F, steps = 24, 7, 160
    bank = FieldBank.from_seed(F, variant, seed=3)
    a = OracleVecEnv(N, bank, variant, radius=radius, bonus=bonus, trend_k=trend_k)
    b = NumpyVecEnv(N, bank, variant, radius=radius, bonus=bonus, trend_k=trend_k)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.RandomState(1)
    ends = 0
    for t in range(steps):
        act = rng.randint(0, 5, N)
        for i, e in enumerate(a.envs):          # half the envs home in on their source (episode ends, resets, boundary)
            if i % 2 == 0:
                d = e.source - e.pos
                act[i] = (3 if d[0] > 0 else 4) if abs(d[0]) > abs(d[1]) else (1 if d[1] > 0 else 2)
        z = rng.randn(N, 2) * (3.0 if t % 7 == 0 else 1.0)
        ra, rb = a.step(act, z), b.step(act, z)
        for x, y, name in zip(ra, rb, ("obs", "rew", "done", "reached", "info", "term")):
            assert np.array_equal(x, y), (t, name)
        ends += int(ra[2].sum())
    assert ends >= 8
    assert np.array_equal(b.episode, a.episode)
