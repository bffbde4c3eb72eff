"""Oracle (CPU restatement) vs the reference's own env traces -- bit-exact.  E1-E5 of SURVEY 8a."""
import numpy as np
import pytest

from oracle.env_oracle import FieldBank, OracleEnv, OracleVecEnv, EnvCore, draw_episode, VARIANTS


@pytest.mark.parametrize("var", ["v2.0", "v2.1", "v1.1"])
def test_env_trace_bit_exact(golden, var):
    g = golden("env_traces.npz")
    env = OracleEnv(var, seed=int(g[f"{var}_seed"]))
    act = g[f"{var}_act"]
    ep = 0
    assert np.array_equal(env.source, g[f"{var}_sources"][0])
    assert np.array_equal(env.obs(), g[f"{var}_obs0"][0])
    for t in range(len(act)):
        o, r, d, s, info = env.step(int(act[t]))
        assert np.array_equal(o, g[f"{var}_obs"][t]), t
        assert r == g[f"{var}_rew"][t], (t, r, g[f"{var}_rew"][t])
        assert d == g[f"{var}_done"][t] and s == g[f"{var}_reached"][t], t
        assert np.array_equal(info, g[f"{var}_info"][t]), t
        assert np.array_equal(env.pos, g[f"{var}_pos"][t]), t
        if g[f"{var}_reset_after"][t]:
            ep += 1
            o0 = env.reset()
            assert np.array_equal(env.source, g[f"{var}_sources"][ep])
            assert np.array_equal(o0, g[f"{var}_obs0"][ep])
            k = min(ep, len(g[f"{var}_curr_radius"]) - 1)
            env.radius = float(g[f"{var}_curr_radius"][k])
            b = g[f"{var}_curr_bonus"][k]
            env.bonus = np.float64(b) if g[f"{var}_curr_bonus_is_f64"][k] else float(b)
    assert ep >= 5 and g[f"{var}_reached"].sum() >= 5


def test_vec_env_equals_single_envs():
    """OracleVecEnv (injected noise, bank of fields, auto-reset) == N independent EnvCore runs."""
    n, F = 3, 6
    bank = FieldBank.from_seed(F, "v2.1", seed=4)
    vec = OracleVecEnv(n, bank, "v2.1", radius=60.0)
    obs = vec.reset()
    singles = [EnvCore("v2.1") for _ in range(n)]
    for i, e in enumerate(singles):
        e.radius = 60.0
        assert np.array_equal(e.begin_episode(bank.sources[i], bank.conc[i], bank.tke[i]), obs[i])
    rng = np.random.RandomState(0)
    epi = [0] * n
    ndone = 0
    for t in range(150):
        # home in on the source so that episodes end and auto-reset is exercised
        act = []
        for e in singles:
            d = e.source - e.pos
            act.append((3 if d[0] > 0 else 4) if abs(d[0]) > abs(d[1]) else (1 if d[1] > 0 else 2))
        z = rng.randn(n, 2)
        obs, rew, done, reached, info, term = vec.step(np.array(act), z)
        for i, e in enumerate(singles):
            o, r, d, s, inf = e.step(act[i], z[i])
            assert r == rew[i] and d == done[i] and np.array_equal(o, term[i])
            if d:
                ndone += 1
                epi[i] += 1
                f = (i + epi[i] * n) % F
                o = e.begin_episode(bank.sources[f], bank.conc[f], bank.tke[f])
            assert np.array_equal(o, obs[i])
    assert ndone >= 3


def test_field_statistics():
    rs = np.random.RandomState(1)
    src, conc, tke = draw_episode(rs, VARIANTS["v2.0"][0])
    assert 50 <= src.min() and src.max() <= 450
    assert conc.min() >= 0 and conc.max() <= 100
    assert abs(tke.mean() - 3 * (np.sqrt(2 / np.pi) + 0.1)) < 0.05
