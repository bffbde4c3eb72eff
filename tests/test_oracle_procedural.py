"""The procedural-mode oracle itself (CPU): Philox4x32-10 against the Random123 known-answer vectors, the cell
formula against `env_oracle.make_fields` (which is bit-exact against the reference's own tables,
tests/test_oracle_env.py), the Box-Muller normals' moments, the inverse-CDF draw."""
import numpy as np

from oracle import env_oracle as eo
from oracle import procedural_oracle as pr


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds: (counter, key) -> output."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = pr.philox4x32_10(key[0] | (key[1] << 32), *ctr)
        assert tuple(int(g) for g in got) == want
    # vectorised == scalar
    idx = np.arange(1000)
    vec = pr.philox4x32_10(77, idx, 3, 9, 2)
    for i in (0, 1, 999):
        assert tuple(int(v[i]) for v in vec) == tuple(int(g) for g in pr.philox4x32_10(77, i, 3, 9, 2))


def test_cell_formula_is_the_reference_pinned_table_formula():
    """field_cells == make_fields (environment.py:51-62 term for term) fed with the counter RNG's draws, bit for bit."""
    seed, env, epi = 1234, 17, 3
    for variant in ("v2.0", "v2.1"):
        sigma = eo.VARIANTS[variant][0]
        src, conc, tke = pr.full_field(seed, env, epi, sigma)
        x, y = np.mgrid[:eo.GRID, :eo.GRID]
        r0, r1, _, r3 = pr.philox4x32_10(seed, x * eo.GRID + y, env, epi, pr.RNG_FIELD)
        gauss = pr._bm_radius(r0) * np.cos(2.0 * np.pi * pr._u24(r1))
        unif = r3.astype(np.float64) / 4294967296.0
        c2, t2 = eo.make_fields(src, sigma, gauss, unif)
        assert np.array_equal(conc, c2) and np.array_equal(tke, t2)
        assert 50.0 <= src.min() and src.max() < 450.0
        # one cell on demand == the table
        lz = pr.LazyField(seed, env, epi, src, sigma, 0)
        assert lz[(123, 45)] == conc[123, 45]


def test_draw_statistics():
    g = pr.step_normals(5, np.arange(200000), 0, 0)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01
    assert abs(np.mean(g[:, 0] * g[:, 1])) < 0.01
    assert abs(np.mean(np.abs(g) > 1.959964) - 0.05) < 0.003
    u = pr.action_uniform(5, 3, np.arange(100000), 0)
    assert u.dtype == np.float32 and 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.005
    s = pr.source_of(5, np.arange(50000), 1)
    assert s.shape == (50000, 2) and abs(s.mean() - 250.0) < 2.0 and s.min() >= 50.0 and s.max() < 450.0


def test_inverse_cdf_draw():
    p = np.array([[0.1, 0.2, 0.3, 0.25, 0.15]] * 5, np.float32)
    u = np.array([0.0, 0.0999, 0.31, 0.86, 0.999], np.float32)
    assert pr.sample_inverse_cdf(p, u).tolist() == [0, 0, 2, 4, 4]


def test_procedural_vec_env_runs_and_auto_resets():
    env = pr.ProceduralVecEnv(6, seed=3, variant="v2.0", radius=400.0)
    obs = env.reset()
    assert obs.shape == (6, 6) and obs.dtype == np.float32
    ended = 0
    for t in range(6):
        obs, rew, done, reached, info, term = env.step(np.full(6, 1 + t % 4))
        ended += int(done.sum())
        assert np.isfinite(rew).all()
    assert ended > 0 and env.episode.sum() == ended
