"""HIP env kernels vs the oracle env (which is bit-exact vs the reference).  -m gpu."""
import numpy as np
import pytest
import torch

from oracle.env_oracle import FieldBank, OracleVecEnv, OracleEnv, EnvCore

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def homing(envs):
    act = []
    for e in envs:
        d = e.source - e.pos
        act.append((3 if d[0] > 0 else 4) if abs(d[0]) > abs(d[1]) else (1 if d[1] > 0 else 2))
    return np.array(act, np.int32)


@pytest.mark.parametrize("variant", ["v2.0", "v2.1", "v1.1"])
def test_materialised_parity_bit_exact(variant):
    """positions / observations / done flags bit-exact, rewards to f64 round-off (pow)."""
    from uavppo.vec_env import VecMethaneEnv
    n, F = 4, 12
    bank = FieldBank.from_seed(F, variant, seed=9)
    ora = OracleVecEnv(n, bank, variant, radius=40.0, bonus=0.6)
    env = VecMethaneEnv(n, variant, DEV, bank=bank.interleaved(), bank_sources=bank.sources)
    env.current_radius = 40.0
    o_ref = ora.reset()
    o = env.reset()
    assert np.array_equal(o.cpu().numpy(), o_ref)
    rng = np.random.RandomState(0)
    n_done = 0
    for t in range(260):
        if t < 120:
            act = homing(ora.envs)
        elif t < 200:
            act = rng.randint(0, 5, n).astype(np.int32)
        else:
            act = np.array([0, 2, 4, 1], np.int32)      # stay / walk into the walls
        if t == 150:   # curriculum change incl. the np.float64 bonus path
            ora.set_curriculum(25.0, np.float64(0.37))
            env.current_radius, env.explore_bonus = 25.0, np.float64(0.37)
        z = rng.randn(n, 2)
        o_ref, r_ref, d_ref, s_ref, info_ref, term_ref = ora.step(act, z)
        o, r, d, info = env.step(torch.from_numpy(act).to(DEV), torch.from_numpy(z).to(DEV))
        assert np.array_equal(o.cpu().numpy(), o_ref), t
        assert np.array_equal(env.term_obs.cpu().numpy(), term_ref), t
        assert np.array_equal(d.cpu().numpy() > 0, d_ref), t
        fl = env.flags.cpu().numpy()
        assert np.array_equal((fl & 1) > 0, d_ref) and np.array_equal((fl & 2) > 0, s_ref), t
        assert np.allclose(env.rew64.cpu().numpy(), r_ref, rtol=0, atol=1e-6), t
        assert np.allclose(r.cpu().numpy(), r_ref.astype(np.float32), rtol=0, atol=1e-6), t
        assert np.allclose(info.cpu().numpy(), info_ref, rtol=0, atol=1e-6), t
        pos, src, steps, epi = env.peek()
        assert np.array_equal(pos.cpu().numpy(), np.stack([e.pos.astype(np.float32) for e in ora.envs])), t
        assert np.array_equal(epi.cpu().numpy(), ora.episode), t
        n_done += int(d_ref.sum())
    assert n_done >= 4


def test_golden_trace_through_kernel(golden):
    """The reference's own V2.0 trace (tests/golden/env_traces.npz) replayed through the HIP step
    kernel, N=1: fields/noise regenerated from the recorded numpy seed by the oracle."""
    from uavppo.vec_env import VecMethaneEnv
    g = golden("env_traces.npz")
    var = "v2.0"
    ora = OracleEnv(var, seed=int(g[f"{var}_seed"]))
    act = g[f"{var}_act"]
    # play the oracle once to collect per-episode tables and per-step noise in draw order
    episodes = [(ora.source.copy(), ora.conc, ora.tke)]
    noise, cur = [], []
    ep = 0
    for t in range(300):
        st = ora.rs.get_state()
        z = ora.rs.randn(2)
        ora.rs.set_state(st)
        noise.append(z)
        cur.append((ora.radius, ora.bonus))
        ora.step(int(act[t]))
        if g[f"{var}_reset_after"][t]:
            ep += 1
            ora.reset()
            k = min(ep, len(g[f"{var}_curr_radius"]) - 1)
            ora.radius = float(g[f"{var}_curr_radius"][k])
            b = g[f"{var}_curr_bonus"][k]
            ora.bonus = np.float64(b) if g[f"{var}_curr_bonus_is_f64"][k] else float(b)
            episodes.append((ora.source.copy(), ora.conc, ora.tke))
    bank = FieldBank(np.stack([e[0] for e in episodes]), np.stack([e[1] for e in episodes]),
                     np.stack([e[2] for e in episodes]))
    env = VecMethaneEnv(1, var, DEV, bank=bank.interleaved(), bank_sources=bank.sources)
    o = env.reset()
    assert np.array_equal(o.cpu().numpy()[0], g[f"{var}_obs0"][0])
    ep = 0
    for t in range(300):
        env.current_radius, env.explore_bonus = cur[t]
        a = torch.tensor([int(act[t])], dtype=torch.int32, device=DEV)
        o, r, d, info = env.step(a, torch.from_numpy(noise[t][None]).to(DEV))
        assert np.array_equal(env.term_obs.cpu().numpy()[0], g[f"{var}_obs"][t]), t
        assert abs(env.rew64.item() - g[f"{var}_rew"][t]) < 1e-6, t
        assert bool(d.item()) == bool(g[f"{var}_done"][t]), t
        if g[f"{var}_reset_after"][t]:
            ep += 1
            if not g[f"{var}_done"][t]:
                break          # the golden script force-reset a non-terminated episode; stop the replay there
            assert np.array_equal(o.cpu().numpy()[0], g[f"{var}_obs0"][ep]), t
    assert ep >= 2


def test_procedural_field_statistics():
    """Procedural mode (counter RNG) is validated statistically: source ~ U[50,450]^2, obs[3] mean
    = E[tke]/9, concentration peaks near the source, per-(env,episode) determinism."""
    from uavppo.vec_env import VecMethaneEnv
    n = 8192
    env = VecMethaneEnv(n, "v2.0", DEV, seed=5)
    o0 = env.reset().cpu().numpy().copy()
    _, src, _, _ = env.peek()
    s = src.cpu().numpy()
    assert s.min() >= 50 and s.max() <= 450 and abs(s.mean() - 250) < 5 and abs(s.std() - 400 / np.sqrt(12)) < 4
    tke_mean = 3 * (np.sqrt(2 / np.pi) + 0.2 * 0.5) + 0.9 * 0.0   # sin(0)=0 at cell (0,0)
    assert abs(o0[:, 3].mean() * 9 - tke_mean) < 0.08
    env2 = VecMethaneEnv(n, "v2.0", DEV, seed=5)
    assert np.array_equal(env2.reset().cpu().numpy(), o0)
    env3 = VecMethaneEnv(n, "v2.0", DEV, seed=6)
    assert not np.array_equal(env3.reset().cpu().numpy(), o0)
    # random walk: rewards finite, steps advance, some boundary contact
    g = torch.Generator(device=DEV).manual_seed(0)
    for t in range(50):
        a = torch.randint(0, 5, (n,), generator=g, device=DEV, dtype=torch.int32)
        o, r, d, info = env.step(a)
    assert torch.isfinite(r).all() and torch.isfinite(o).all()
    assert (o[:, 4] >= 0).all() and float(o[:, 4].max()) == pytest.approx(0.05) and float(o[:, 0].max()) <= 0.998 + 1e-6
    _, _, steps, _ = env.peek()
    assert int(steps.max()) == 50


def test_env_offset_shards_are_consistent():
    """Rank sharding: envs [4,8) of an 8-env job == a 4-env shard with env_offset=4 (no collective)."""
    from uavppo.vec_env import VecMethaneEnv
    full = VecMethaneEnv(8, "v2.1", DEV, seed=3)
    shard = VecMethaneEnv(4, "v2.1", DEV, seed=3, env_offset=4, n_env_total=8)
    assert np.array_equal(full.reset().cpu().numpy()[4:], shard.reset().cpu().numpy())
    a = torch.tensor([1, 3, 1, 3, 1, 3, 1, 3], dtype=torch.int32, device=DEV)
    for _ in range(20):
        of, rf, _, _ = full.step(a)
        os_, rs, _, _ = shard.step(a[4:].contiguous())
        assert torch.equal(of[4:], os_) and torch.equal(rf[4:], rs)


@pytest.mark.parametrize("k", [1, 2])
def test_trend_observation_channels(k):
    """BASELINE C5 'trend obs': obs[6+i] = obs[2](t) - obs[2](t-1-i); the 6 reference features are unchanged
    (bit-exact) and the extra channels equal the oracle's f32 differences, across auto-resets."""
    from uavppo.vec_env import VecMethaneEnv
    n, F = 3, 9
    bank = FieldBank.from_seed(F, "v2.1", seed=12)
    ora = OracleVecEnv(n, bank, "v2.1", radius=55.0, trend_k=k)
    env = VecMethaneEnv(n, "v2.1", DEV, bank=bank.interleaved(), bank_sources=bank.sources, trend_k=k)
    base = VecMethaneEnv(n, "v2.1", DEV, bank=bank.interleaved(), bank_sources=bank.sources)
    env.current_radius = base.current_radius = 55.0
    o_ref = ora.reset()
    assert env.obs_dim == 6 + k and np.array_equal(env.reset().cpu().numpy(), o_ref)
    base.reset()
    rng = np.random.RandomState(5)
    ndone = 0
    for t in range(140):
        act = homing(ora.envs) if t % 3 else rng.randint(0, 5, n).astype(np.int32)
        z = rng.randn(n, 2)
        o_ref, r_ref, d_ref, _, _, term_ref = ora.step(act, z)
        a, zz = torch.from_numpy(act).to(DEV), torch.from_numpy(z).to(DEV)
        o, r, d, _ = env.step(a, zz)
        ob, rb, _, _ = base.step(a, zz)
        assert np.array_equal(o.cpu().numpy(), o_ref), t
        assert np.array_equal(env.term_obs.cpu().numpy(), term_ref), t
        assert torch.equal(o[:, :6], ob) and torch.equal(r, rb)          # reference features / rewards untouched
        ndone += int(d_ref.sum())
    assert ndone >= 2
