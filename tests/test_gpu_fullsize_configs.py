"""BASELINE.json's OTHER configurations at their exact per-GPU shapes (C3 is tests/test_gpu_fullsize.py):

  C2  256 envs x 64 steps, LSTM h=64                      whole iteration against the oracle (16,384 samples: CPU seconds)
  C4  1024 envs x 128 steps per GPU, h=128, PPOV2.1 (sigma 15), materialised bank of F=64 fields built by
      uav_env_materialise, through rollout_lstm_kernel    oracle rows + invariants + whole iteration against the oracle
  C5  4096 envs x 256 steps per GPU, h=256 x 2, obs 6+2   tile independence, determinism, fp16-split vs exact-f32
      gradient, pipelined vs per-layer backward bit-equality -- properties at 1 M samples; ONE step-kernel tile of it
      (64 envs x 256 steps, the same kernels) whole iteration against the oracle

Reference shapes: PPOV2.1/environment.py:52-69 (sigma = 15 field), nn.LSTM stack PPOV2.0/model.py:206-212,
update loop PPOV2.0/train_ppo2.0.py:15-88.  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import procedural_oracle as pr
from oracle.env_oracle import EnvCore

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cpu_params(policy):
    return {k: v.detach().cpu().clone() for k, v in policy.named_views().items()}


# ------------------------------------------------------------------------------------------------------------ C2
def test_c2_whole_iteration_matches_oracle():
    """One full iteration at C2's exact shape: procedural rollout (own actions, own noise) replayed by the oracle, then
    GAE + normalise + 5 full-batch epochs against the oracle's torch-autograd update on the GPU's buffers: per-epoch
    losses, clipped gradient norms and the parameters after the fifth Adam step."""
    from uavppo.trainer import VecPPOTrainer
    from test_gpu_procedural import _check_rollout_against_oracle
    from test_gpu_trainer import oracle_lstm_update
    N, T, H, seed = 256, 64, 64, 2025
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, variant="v2.0", device=DEV, seed=seed, use_curriculum=False)
    tr.radius = 120.0
    tr.reset()
    ora = pr.ProceduralVecEnv(N, seed, "v2.0", radius=120.0)
    obs0 = ora.reset()
    p = cpu_params(tr.policy)
    adam = po.AdamState(p)
    h0, c0 = tr.h.cpu().clone(), tr.c.cpu().clone()
    tr.collect()
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    assert _check_rollout_against_oracle(tr, ora, obs0, b) >= 10          # episodes end inside the rollout
    tr.record = True
    tr.update()
    log, adv_n, ret = oracle_lstm_update(p, adam, b["obs"], b["act"], b["rew"], b["val"], b["logp"], b["done"], b["keep"],
                                         h0, c0, tr.hp["epochs"], "reference_exact", None)
    assert np.allclose(tr.adv_n.cpu().numpy().reshape(-1), adv_n, atol=3e-5, rtol=1e-4)
    assert np.allclose(tr.ret.cpu().numpy().reshape(-1), ret, atol=3e-5, rtol=1e-4)
    assert len(tr.log) == 5
    for e, (sums, gn) in enumerate(tr.log):
        got = sums.cpu().numpy()[:3] / (N * T)
        assert np.allclose(got, log[e, :3], rtol=2e-4, atol=2e-6), (e, got, log[e])
        assert np.isclose(gn.item(), log[e, 3], rtol=2e-3), (e, gn.item(), log[e, 3])
    # parameters after the fifth Adam step of the reference's lr (3e-5): the bar of the small-shape update tests
    # (tests/test_gpu_trainer.py: 3e-6 = a tenth of one step) -- Adam turns a gradient element's relative error into
    # lr x that error, and elements that nearly cancel over 16,384 samples carry the largest
    # A few elements are exempt by construction: where the gradient cancels to rounding noise -- the critic bias at epoch 0
    # is sum(V - ret) = -sum(adv_n) = 0 by the normalisation itself -- Adam's step is lr * sign(noise); such an element
    # may differ by up to 2 * epochs * lr.
    got = cpu_params(tr.policy)
    n_far = n_all = 0
    for k in p:
        d = (got[k] - p[k]).abs()
        n_far += int((d > 3e-6).sum())
        n_all += d.numel()
        assert d.max().item() <= 2 * 5 * 3e-5, (k, d.max().item())
    assert n_far <= 1e-3 * n_all + 2, (n_far, n_all)


# ------------------------------------------------------------------------------------------------------------ C4
@pytest.fixture(scope="module")
def c4():
    """C4's per-GPU trainer exactly as bench.py builds it: bank of 64 sigma=15 fields from uav_env_materialise."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    from uavppo.vec_env import VecMethaneEnv
    F, N, T = 64, 1024, 128
    gen = VecMethaneEnv(F, "v2.1", DEV, seed=4321)
    gen.reset()
    bank = torch.stack([ops.env_materialise(gen.state, F, gen.cfg(), f) for f in range(F)])
    src = gen.peek()[1]
    tr = VecPPOTrainer(N, T, "lstm", hidden=128, variant="v2.1", device=DEV, seed=1234, bank=bank, bank_sources=src,
                       use_curriculum=False)
    tr.radius = 90.0
    tr.reset()
    obs0 = tr.cur_obs.cpu().numpy().copy()
    tr.collect()
    return tr, bank.cpu().numpy(), src.cpu().numpy(), obs0


def test_c4_bank_is_the_sigma15_field(c4):
    """The bank bench.py's C4 uses IS E3 with sigma = 15 (PPOV2.1/environment.py:56): cell for cell against the oracle."""
    _, bank, src, _ = c4
    for f in (0, 17, 63):
        want_src, conc, tke = pr.full_field(4321, f, 0, 15.0)
        assert np.array_equal(src[f], want_src)
        assert np.abs(bank[f, ..., 0] - conc).max() <= 1e-11 and np.abs(bank[f, ..., 1] - tke).max() <= 1e-11
    # a sigma = 15 plume is narrow: 100 ppm at the source, below 1 + turbulence 60 cells away
    f = 5
    sx, sy = int(src[f, 0]), int(src[f, 1])
    assert bank[f, sx, sy, 0] == 100.0 and bank[f, min(sx + 60, 499), sy, 0] < 20.0


def test_c4_rollout_rows_equal_oracle(c4):
    """rollout_lstm_kernel with variant v2.1 over the materialised bank at 1024 x 128: 16 env rows (first, last, and
    rows whose episodes ended) replayed by the oracle from the bank tables -- field choice (env + episode * N) mod F,
    the counter RNG's step noise -- bit for bit; whole-buffer invariants."""
    tr, bank, src, obs0 = c4
    N, T, F = tr.N, tr.T, bank.shape[0]
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    ended = np.nonzero(b["done"].sum(1) > 0)[0]
    assert len(ended) >= 8
    rows = sorted(set([0, 1, 15, 16, N - 1] + ended[:11].tolist()))
    for i in rows:
        e = EnvCore("v2.1")
        e.radius = 90.0
        episode = 0

        def begin():
            f = (i + episode * N) % F
            return e.begin_episode(src[f], bank[f, ..., 0], bank[f, ..., 1])

        o = begin()
        assert np.array_equal(o, obs0[i])
        for t in range(T):
            assert np.array_equal(b["obs"][i, t], o), (i, t)
            z = pr.step_normals(1234, i, episode, e.steps)
            o, r, d, reached, _ = e.step(int(b["act"][i, t]), z)
            assert abs(b["rew"][i, t] - np.float32(r)) <= 1e-6 and bool(b["done"][i, t]) == d, (i, t)
            assert int(b["flags"][i, t]) == int(d) + 2 * int(reached), (i, t)
            if d:
                episode += 1
                o = begin()
    # invariants over all 131,072 samples
    bt = tr.buf
    assert torch.isfinite(bt["rew"]).all() and torch.isfinite(bt["val"]).all() and tr.nan_count.item() == 0
    assert torch.equal(bt["keep"][:, 1:], 1 - bt["done"][:, :-1]) and bool((bt["keep"][:, 0] == 1).all())
    assert int(bt["act"].min()) >= 0 and int(bt["act"].max()) <= 4
    # the stash the rollout emitted is the forward pass of the same parameters (what PPO epoch 0 adopts)
    from uavppo import ops
    v = tr.policy.views
    y, _, _, _ = ops.lstm_fwd(bt["obs"][:48].contiguous(), bt["keep"][:48].contiguous(), tr.h0[0][:48].contiguous(),
                              tr.c0[0][:48].contiguous(), v["lstm.weight_ih_l0"], v["lstm.weight_hh_l0"],
                              v["lstm.bias_ih_l0"], v["lstm.bias_hh_l0"])
    assert torch.allclose(tr.work["y0"][:48], y, atol=3e-6)


def test_c4_iteration_is_deterministic(c4):
    from uavppo.trainer import VecPPOTrainer
    tr, bank, src, _ = c4
    outs = []
    for _ in range(2):
        t2 = VecPPOTrainer(tr.N, tr.T, "lstm", hidden=128, variant="v2.1", device=DEV, seed=1234, bank=tr.bank,
                           bank_sources=tr.bank_sources, use_curriculum=False, epochs=2)
        t2.radius = 90.0
        t2.reset()
        t2.train_iteration()
        outs.append((t2.policy.flat.clone(), t2.loss_sums.clone(), t2.buf["obs"].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], tr.buf["obs"])           # and the same rollout as the fixture's


def test_c4_whole_iteration_matches_oracle(c4):
    """C4's per-GPU iteration (1024 x 128, h = 128, sigma = 15 bank, radius 90 so that episodes end inside the rollout):
    update on the rollout's own buffers against the oracle update -- losses, gradients, norms, parameters."""
    from _iteration_check import update_vs_oracle
    from uavppo.trainer import VecPPOTrainer
    tr, _, _, _ = c4
    t2 = VecPPOTrainer(tr.N, tr.T, "lstm", hidden=128, variant="v2.1", device=DEV, seed=1234, bank=tr.bank,
                       bank_sources=tr.bank_sources, use_curriculum=False)
    t2.radius = 90.0
    t2.reset()
    t2.collect()
    assert torch.equal(t2.buf["obs"], tr.buf["obs"])         # the rollout the row-replay test pinned to the oracle
    m = update_vs_oracle(t2, "c4")
    assert m["samples"] == 131072 and m["episode_ends"] >= 8


# ------------------------------------------------------------------------------------------------------------ C5
C5 = dict(N=4096, T=256, H=256, L=2, K=2)


def test_c5_tile_whole_iteration_matches_oracle():
    """One tile of C5 -- 64 envs x 256 steps, h = 256 x 2, TREND_K = 2, radius 150 (restarts inside the sequence) -- through the
    SAME kernels as the full shape (stepper rollout whose stash epoch 0 adopts, step_fwd_h3 / cell_bwd_h3 / step_bwd_h3 with
    the pipelined two-layer backward, split-fp16 weight-gradient GEMMs, colsum_xw): rollout replayed by the procedural oracle,
    then the whole update against the oracle's two-layer torch-CPU LSTM (nn.LSTM stack semantics, PPOV2.0/model.py:206-212)."""
    from _iteration_check import update_vs_oracle
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    N, T = 64, C5["T"]
    tr = VecPPOTrainer(N, T, "lstm", hidden=C5["H"], layers=C5["L"], variant="v2.1", device=DEV, seed=1234,
                       use_curriculum=False, trend_k=C5["K"])
    tr.radius = 150.0
    tr.reset()
    ora = pr.ProceduralVecEnv(N, 1234, "v2.1", radius=150.0, trend_k=C5["K"])
    obs = ora.reset()
    tr.collect()
    assert tr._rollout_forward_valid and ops.lstm_bwd_caps(DEV, 6 + C5["K"], 256) != 0      # the fp16-split step path
    b = {k: tr.buf[k].cpu().numpy() for k in ("obs", "act", "rew", "done")}
    for t in range(T):
        assert np.array_equal(b["obs"][:, t], obs), t
        obs, rew, done, _, _, _ = ora.step(b["act"][:, t])
        assert np.allclose(b["rew"][:, t], rew.astype(np.float32), atol=1e-6, rtol=0) and np.array_equal(b["done"][:, t] > 0, done)
    m = update_vs_oracle(tr, "c5_tile")
    assert m["samples"] == 16384 and m["episode_ends"] >= 4


@pytest.fixture(scope="module")
def c5():
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(C5["N"], C5["T"], "lstm", hidden=C5["H"], layers=C5["L"], variant="v2.1", device=DEV, seed=1234,
                       use_curriculum=False, trend_k=C5["K"])
    tr.radius = 150.0
    tr.reset()
    tr.collect()
    tr.compute_advantages()
    tr.rollout_heads = tr.work["heads"].clone()          # what the stepper rollout wrote (pol.heads() reuses the buffer)
    yield tr
    del tr
    torch.cuda.empty_cache()


def _c5_gradient(tr, arith=None, stack=True):
    """PPO gradient of the collected buffers: forward over the stored observations, loss, backward."""
    from uavppo import ops
    b, pol = tr.buf, tr.policy
    n = tr.N * tr.T
    args = (b["act"].reshape(-1), b["logp"].reshape(-1), tr.adv_n.reshape(-1), tr.ret.reshape(-1), b["val"].reshape(-1),
            1.0 / n, 0.2, 0.01)
    ops.set_lstm_arith(arith or "fp16x3")
    pol.use_stack_bwd = stack
    try:
        heads = pol.heads(b["obs"], b["keep"], tr.h0, tr.c0, tr.work)
        loss = torch.zeros(4, dtype=torch.float64, device=DEV)
        dheads = torch.empty(n, 6, device=DEV)
        dbias = torch.empty(6, device=DEV)
        ops.ppo_loss_heads(heads.view(n, -1), *args, loss, dheads, dbias)
        g = pol.backward(dheads, tr.work, dbias).clone()
    finally:
        ops.set_lstm_arith("fp16x3")
        pol.use_stack_bwd = True
    return g, loss, heads.clone()


def test_c5_rollout_invariants_and_tile_independence(c5):
    """The stepper rollout at 4096 x 256: invariants over the 1 M samples; the first 64 envs (one tile of the step
    kernels) equal, bit for bit, a 64-env job with the same seed (draws are keyed by the GLOBAL env index, tiles are
    independent problems); the rollout's stash IS the update's forward pass (epoch 0 adopts it)."""
    from uavppo.trainer import VecPPOTrainer
    tr = c5
    b = tr.buf
    assert torch.isfinite(b["rew"]).all() and torch.isfinite(b["val"]).all() and tr.nan_count.item() == 0
    assert torch.equal(b["keep"][:, 1:], 1 - b["done"][:, :-1]) and bool((b["keep"][:, 0] == 1).all())
    assert int(b["done"].sum()) > 100
    # trend channels: obs[6] = obs[2](t) - obs[2](t-1) inside an episode
    o = b["obs"]
    same = b["done"][:, :-1] == 0
    assert torch.equal(o[:, 1:, 6][same], (o[:, 1:, 2] - o[:, :-1, 2])[same])
    small = VecPPOTrainer(64, C5["T"], "lstm", hidden=C5["H"], layers=C5["L"], variant="v2.1", device=DEV, seed=1234,
                          use_curriculum=False, trend_k=C5["K"])
    small.radius = 150.0
    small.reset()
    small.collect()
    for k in ("obs", "act", "rew", "val", "logp", "done", "keep"):
        assert torch.equal(small.buf[k], b[k][:64]), k
    for l in range(C5["L"]):
        assert torch.equal(small.work[f"y{l}"], tr.work[f"y{l}"][:64]), l
    # oracle rows: three envs replayed from the procedural oracle (8 observation channels)
    ora = pr.ProceduralVecEnv(3, 1234, "v2.1", radius=150.0, trend_k=2)
    obs = ora.reset()
    bn = {k: b[k][:3].cpu().numpy() for k in ("obs", "act", "rew", "done")}
    for t in range(C5["T"]):
        assert np.array_equal(bn["obs"][:, t], obs), t
        obs, rew, done, _, _, _ = ora.step(bn["act"][:, t])
        assert np.allclose(bn["rew"][:, t], rew.astype(np.float32), atol=1e-6, rtol=0) and np.array_equal(bn["done"][:, t] > 0, done)


def test_c5_gradient_fp16_split_equals_exact_f32_and_pipelined_equals_per_layer(c5):
    """At the full C5 shape: (a) the whole-batch PPO gradient of the default path (fp16-split step kernels, split
    GEMMs, pipelined two-layer backward) equals the exact-f32 path's to f32 summation noise; (b) the pipelined backward
    (uav_lstm_bwd_stack) equals the per-layer calls BIT for bit; (c) recomputing is deterministic; (d) the heads of the
    recomputed forward equal the ones the stepper rollout wrote."""
    tr = c5
    g_stack, loss_a, heads_a = _c5_gradient(tr)
    assert torch.equal(heads_a.view(-1), tr.rollout_heads.view(-1))                           # (d)
    g_again, loss_b, _ = _c5_gradient(tr)
    assert torch.equal(g_stack, g_again) and torch.equal(loss_a, loss_b)                      # (c)
    g_layers, _, _ = _c5_gradient(tr, stack=False)
    assert torch.equal(g_stack, g_layers)                                                     # (b)
    g_f32, loss_f, heads_f = _c5_gradient(tr, arith="f32_mfma")
    assert torch.isfinite(g_f32).all()
    rel = ((g_stack.double() - g_f32.double()).norm() / g_f32.double().norm()).item()
    assert rel < 3e-5, rel                                                                    # (a)
    assert torch.allclose(heads_a, heads_f, atol=2e-5, rtol=1e-4)
    assert torch.allclose(loss_a[1:3], loss_f[1:3], rtol=1e-6)
