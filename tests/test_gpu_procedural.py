"""Procedural-field mode (the mode bench.py measures) against oracle/procedural_oracle.py: Philox keying, the E3 cell
formula of PPOV2.0/environment.py:51-62 (PPOV2.1/environment.py:52-61) in f64 given the draws, the Box-Muller step
noise of environment.py:101, the action uniform of train_ppo2.0.py:162.  Only the random STREAM (Philox instead of
numpy's MT19937) is the product's own; everything computed from the draws is pinned here.  -m gpu.

Stated bounds: field cells |d conc|, |d tke| <= 1e-11 on the 0..100 scale (libm exp / log / cos against numpy's, a few
ulp); observations therefore BIT-equal except where an f32 rounding boundary falls inside that 1e-13 relative band
(probability ~1e-6 per value: a handful of 1-ulp differences are tolerated, none observed), done flags exact,
rewards 1e-6 (the f64 `pow` table, as in materialised mode)."""
import numpy as np
import pytest
import torch

from oracle import procedural_oracle as pr
from oracle import ppo_oracle as po
from oracle.env_oracle import VARIANTS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIELD_ATOL = 1e-11


def _assert_obs_equal(got, want, what, max_ulp_cells=3):
    """bit-equal; a few 1-ulp f32 differences allowed (see the module docstring), anything larger fails"""
    if np.array_equal(got, want):
        return
    bad = got != want
    assert bad.sum() <= max_ulp_cells, (what, int(bad.sum()))
    assert np.all(np.abs(got[bad] - want[bad]) <= np.spacing(np.abs(want[bad]).astype(np.float32))), what


@pytest.mark.parametrize("variant", ["v2.0", "v2.1", "v1.1"])
def test_materialised_procedural_field_equals_oracle_cell_for_cell(variant):
    """uav_env_materialise (= field_at of every cell, the function the rollouts call) against the f64 formula:
    3 envs x 2 episodes per variant, with a shard offset so the GLOBAL env index is what keys the draws."""
    from uavppo import ops
    from uavppo.vec_env import VecMethaneEnv
    n, off, seed = 6, 40, 20250 + len(variant)
    sigma = VARIANTS[variant][0]
    env = VecMethaneEnv(n, variant, DEV, seed=seed, env_offset=off, n_env_total=64)
    env.current_radius = 2000.0                 # every first step reaches the source: episode 1 starts at once
    obs0 = env.reset().cpu().numpy()
    for episode in (0, 1):
        if episode == 1:
            env.step(torch.ones(n, dtype=torch.int32, device=DEV))
        _, src, steps, epi = env.peek()
        assert epi.cpu().tolist() == [episode] * n and steps.cpu().tolist() == [0] * n
        for i in (0, 2, 5):
            want_src, conc, tke = pr.full_field(seed, off + i, episode, sigma)
            assert np.array_equal(src[i].cpu().numpy(), want_src)             # source: exact (53-bit uniforms)
            got = ops.env_materialise(env.state, n, env.cfg(), i).cpu().numpy()
            assert np.abs(got[..., 0] - conc).max() <= FIELD_ATOL, (variant, episode, i)
            assert np.abs(got[..., 1] - tke).max() <= FIELD_ATOL, (variant, episode, i)
            # what the policy sees of it (obs[2], obs[3]) is bit-equal on (all but possibly a few of) the 250,000 cells
            _assert_obs_equal((got[..., 0] / 100.0).astype(np.float32), (conc / 100.0).astype(np.float32), "obs2")
            _assert_obs_equal((got[..., 1] / 9.0).astype(np.float32), (tke / 9.0).astype(np.float32), "obs3")
            assert conc.max() == 100.0 and abs(np.unravel_index(conc.argmax(), conc.shape)[0] - want_src[0]) < 40
    # reset observation of episode 0 = cell (0,0) of those fields
    want0 = pr.ProceduralVecEnv(n, seed, variant, env_offset=off).reset()
    _assert_obs_equal(obs0, want0, "obs0")


@pytest.mark.parametrize("variant,radius", [("v2.0", 50.0), ("v2.1", 160.0)])
def test_step_kernel_with_its_own_noise_equals_oracle(variant, radius):
    """uav_env_step with noise = NULL: field lookups AND the counter RNG's Box-Muller step noise against the oracle,
    with episode ends (auto-reset -> episode counter keys the next field)."""
    from uavppo.vec_env import VecMethaneEnv
    n, seed, off = 48, 99, 1000
    env = VecMethaneEnv(n, variant, DEV, seed=seed, env_offset=off, n_env_total=4096)
    ora = pr.ProceduralVecEnv(n, seed, variant, radius=radius, env_offset=off)
    env.current_radius = radius
    _assert_obs_equal(env.reset().cpu().numpy(), ora.reset(), "reset")
    rng = np.random.RandomState(1)
    ndone = 0
    for t in range(90):
        act = rng.randint(0, 5, n).astype(np.int32) if t % 4 == 0 else np.where(np.arange(n) % 2, 1, 3).astype(np.int32)
        o_ref, r_ref, d_ref, s_ref, info_ref, term_ref = ora.step(act)
        o, r, d, info = env.step(torch.from_numpy(act).to(DEV))
        _assert_obs_equal(o.cpu().numpy(), o_ref, t)
        _assert_obs_equal(env.term_obs.cpu().numpy(), term_ref, t)
        assert np.array_equal(d.cpu().numpy() > 0, d_ref), t
        assert np.array_equal((env.flags.cpu().numpy() & 2) > 0, s_ref), t
        assert np.allclose(env.rew64.cpu().numpy(), r_ref, rtol=0, atol=1e-6), t
        assert np.allclose(info.cpu().numpy(), info_ref, rtol=0, atol=1e-6), t
        pos, _, _, epi = env.peek()
        assert np.array_equal(pos.cpu().numpy(), np.stack([e.pos.astype(np.float32) for e in ora.envs])), t
        assert np.array_equal(epi.cpu().numpy(), ora.episode), t
        ndone += int(d_ref.sum())
    assert ndone >= 3


def _check_rollout_against_oracle(tr, ora, obs_first, b):
    N, T = tr.N, tr.T
    want = {k: np.zeros_like(b[k]) for k in ("obs", "rew", "done", "keep")}
    flags = np.zeros((N, T), np.uint8)
    obs, keep = obs_first, np.ones(N, np.float32)
    for t in range(T):
        want["obs"][:, t] = obs
        want["keep"][:, t] = keep
        obs, rew, done, reached, _, _ = ora.step(b["act"][:, t])
        want["rew"][:, t] = rew.astype(np.float32)
        want["done"][:, t] = done
        flags[:, t] = done.astype(np.uint8) | (reached.astype(np.uint8) << 1)
        keep = 1.0 - done.astype(np.float32)
    _assert_obs_equal(b["obs"], want["obs"], "obs")
    assert np.array_equal(b["done"], want["done"])
    assert np.array_equal(b["keep"], want["keep"])
    assert np.array_equal(b["flags"], flags)
    assert np.allclose(b["rew"], want["rew"], rtol=0, atol=1e-6)
    _assert_obs_equal(tr.cur_obs.cpu().numpy(), obs, "cur_obs")
    return int(want["done"].sum())


def _check_sampling(heads, act, seed, iteration, off):
    """The rollout's Categorical draw: action == inverse CDF of softmax(logits) at the counter RNG's uniform.  The
    device's softmax uses v_exp_f32, numpy's exp differs in the last bits, so draws whose uniform falls within 2e-6 of
    a CDF edge are not decidable from here and are skipped (a vanishing fraction)."""
    N, T = act.shape
    logits = heads[..., :5].astype(np.float32)
    z = logits - logits.max(-1, keepdims=True)
    e = np.exp(z, dtype=np.float32)
    p = e / e.sum(-1, keepdims=True, dtype=np.float32)
    u = np.stack([pr.action_uniform(seed, t, off + np.arange(N), iteration) for t in range(T)], axis=1)
    cdf = np.cumsum(p, -1, dtype=np.float32)
    target = u * cdf[..., -1]
    clear = np.all(np.abs(cdf - target[..., None]) > 2e-6, axis=-1)
    sel = pr.sample_inverse_cdf(p.reshape(-1, 5), u.reshape(-1)).reshape(N, T)
    assert clear.mean() > 0.999
    assert np.array_equal(sel[clear], act[clear])
    return p


@pytest.mark.parametrize("variant,radius", [("v2.0", 50.0), ("v2.0", 140.0), ("v2.1", 140.0)])
def test_fused_lstm_rollout_procedural_c2_shape_equals_oracle(variant, radius):
    """BASELINE C2's exact shape (256 envs x 64 steps, LSTM h=64), procedural fields, NO injected noise and NO forced
    actions -- the configuration bench.py runs: the oracle consumes the GPU's sampled actions and reproduces every
    observation / done / keep / flag bit for bit and the rewards to 1e-6; the sampled actions themselves are the
    inverse-CDF draw of the recorded logits at the oracle's Philox uniform; values / log-probs match the f32 oracle
    LSTM on the recorded observations.  Second rollout (iteration 1) continues the same episodes."""
    from uavppo.trainer import VecPPOTrainer
    N, T, H, seed = 256, 64, 64, 4242
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, variant=variant, device=DEV, seed=seed, use_curriculum=False)
    tr.radius = radius
    tr.reset()
    ora = pr.ProceduralVecEnv(N, seed, variant, radius=radius)
    obs = ora.reset()
    _assert_obs_equal(tr.cur_obs.cpu().numpy(), obs, "reset")
    p = {k: v.detach().cpu().clone() for k, v in tr.policy.named_views().items()}
    ndone = 0
    for it in range(2):
        tr.iteration = it
        h0, c0 = tr.h.cpu().clone(), tr.c.cpu().clone()
        tr.collect()
        b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
        ndone += _check_rollout_against_oracle(tr, ora, obs, b)
        obs = tr.cur_obs.cpu().numpy().copy()
        heads = tr.work["heads"].cpu().numpy()
        probs_dev = _check_sampling(heads, b["act"], seed, it, 0)
        # policy side on the recorded inputs: f32 oracle LSTM (torch CPU)
        with torch.no_grad():
            probs, value, _, _ = po.lstm_policy_forward(p, torch.from_numpy(b["obs"]).transpose(0, 1), h0, c0,
                                                        keep=torch.from_numpy(b["keep"]).transpose(0, 1))
            lp = po.categorical_logp(probs.transpose(0, 1).reshape(N * T, -1), torch.from_numpy(b["act"]).reshape(-1).long())
        assert np.allclose(b["val"], value.transpose(0, 1).numpy().reshape(N, T), atol=2e-5, rtol=1e-4)
        assert np.allclose(b["logp"], lp.numpy().reshape(N, T), atol=2e-5, rtol=1e-4)
        assert np.allclose(probs_dev, probs.transpose(0, 1).numpy(), atol=2e-6)
        assert tr.nan_count.item() == 0
    if radius > 100:
        assert ndone >= 20          # episode ends (and therefore episode-1 / episode-2 fields) are exercised
    print(f"episodes ended: {ndone}")


def test_fused_mlp_rollout_procedural_equals_oracle():
    """The reference's MLP policy through rollout_mlp_kernel in procedural mode (same env core, own kernel)."""
    from uavppo.trainer import VecPPOTrainer
    N, T, seed = 96, 48, 77
    tr = VecPPOTrainer(N, T, "mlp", variant="v2.0", device=DEV, seed=seed, use_curriculum=False, rank=1, world_size=2)
    tr.radius = 150.0
    tr.reset()
    ora = pr.ProceduralVecEnv(N, seed, "v2.0", radius=150.0, env_offset=N)      # rank 1 of 2: global envs [N, 2N)
    obs = ora.reset()
    _assert_obs_equal(tr.cur_obs.cpu().numpy(), obs, "reset")
    tr.iteration = 5
    tr.collect()
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    b["keep"] = np.ones((N, T), np.float32)
    b["keep"][:, 1:] = 1.0 - b["done"][:, :-1]
    ndone = _check_rollout_against_oracle(tr, ora, obs, b)
    assert ndone >= 5
    # sampled actions: inverse CDF of the oracle MLP's probabilities (f32 torch) at the Philox uniform
    p = {k: v.detach().cpu().clone() for k, v in tr.policy.named_views().items()}
    with torch.no_grad():
        probs, value, _ = po.mlp_forward(p, torch.from_numpy(b["obs"]).reshape(N * T, 6))
    u = np.stack([pr.action_uniform(seed, t, N + np.arange(N), 5) for t in range(T)], axis=1)
    pn = probs.numpy().reshape(N, T, 5)
    cdf = np.cumsum(pn, -1, dtype=np.float32)
    clear = np.all(np.abs(cdf - (u * cdf[..., -1])[..., None]) > 5e-6, axis=-1)
    sel = pr.sample_inverse_cdf(pn.reshape(-1, 5), u.reshape(-1)).reshape(N, T)
    assert clear.mean() > 0.999 and np.array_equal(sel[clear], b["act"][clear])
    assert np.allclose(b["val"], value.numpy().reshape(N, T), atol=2e-5, rtol=1e-4)


def test_stepwise_stacked_rollout_procedural_equals_oracle():
    """C5's rollout path (h=256 x 2 + two trend channels: stepper + uav_policy_sample_at + uav_env_step), procedural."""
    from uavppo.trainer import VecPPOTrainer
    N, T, seed = 64, 24, 11
    tr = VecPPOTrainer(N, T, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=seed, use_curriculum=False, trend_k=2)
    tr.radius = 150.0
    tr.reset()
    ora = pr.ProceduralVecEnv(N, seed, "v2.1", radius=150.0, trend_k=2)
    obs = ora.reset()
    _assert_obs_equal(tr.cur_obs.cpu().numpy(), obs, "reset")
    tr.collect()
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    ndone = _check_rollout_against_oracle(tr, ora, obs, b)
    assert ndone >= 2
    _check_sampling(tr.work["heads"].cpu().numpy(), b["act"], seed, 0, 0)
