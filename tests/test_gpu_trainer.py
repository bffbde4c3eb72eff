"""Trainer-level parity on the GPU: fused rollout vs the oracle simulation, LSTM/MLP PPO update vs
the oracle update, and the reference's 24-update N=1 loss curve (tests/golden/e2e_v20.npz).  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle.env_oracle import FieldBank, OracleEnv, OracleVecEnv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cpu_params(policy):
    return {k: v.detach().cpu().clone() for k, v in policy.named_views().items()}


# --------------------------------------------------------------------------------------------- rollout
@pytest.mark.parametrize("H,N,T", [(64, 5, 40), (128, 19, 70)])
def test_fused_rollout_matches_oracle_simulation(H, N, T):
    """Injected noise + forced actions + materialised bank: every stored quantity of the fused
    persistent rollout equals a step-by-step oracle simulation (env bit-exact, policy to f32 tol)."""
    from uavppo.trainer import VecPPOTrainer
    F = 3 * N
    bank = FieldBank.from_seed(F, "v2.0", seed=31)
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, variant="v2.0", device=DEV, seed=5, bank=bank.interleaved(),
                       bank_sources=bank.sources, gae_mode="standard", use_curriculum=False)
    tr.radius = 45.0
    tr.reset()
    rng = np.random.RandomState(2)
    noise = rng.randn(N, T, 2)
    ora = OracleVecEnv(N, bank, "v2.0", radius=45.0)
    obs = ora.reset()
    assert np.array_equal(tr.cur_obs.cpu().numpy(), obs)
    p = cpu_params(tr.policy)
    h = torch.zeros(1, N, H)
    c = torch.zeros(1, N, H)
    # scripted actions: home in on the source for a while (forces episode ends), then random
    acts = np.zeros((N, T), np.int32)
    want = {k: [] for k in ("obs", "rew", "done", "val", "logp", "keep")}
    keep = np.ones(N, np.float32)
    for t in range(T):
        a = []
        for i, e in enumerate(ora.envs):
            d = e.source - e.pos
            hom = (3 if d[0] > 0 else 4) if abs(d[0]) > abs(d[1]) else (1 if d[1] > 0 else 2)
            a.append(hom if (t < 30 or i % 2 == 0) else int(rng.randint(0, 5)))
        acts[:, t] = a
        with torch.no_grad():
            k = torch.from_numpy(keep)[None]
            probs, value, _, (h, c) = po.lstm_policy_forward(p, torch.from_numpy(obs)[None], h, c, keep=k)
            lp = po.categorical_logp(probs[0], torch.tensor(a))
        want["obs"].append(obs.copy())
        want["val"].append(value[0].numpy().copy())
        want["logp"].append(lp.numpy().copy())
        want["keep"].append(keep.copy())
        obs, rew, done, reached, info, term = ora.step(np.array(a), noise[:, t])
        want["rew"].append(rew.astype(np.float32))
        want["done"].append(done.astype(np.float32))
        keep = 1.0 - done.astype(np.float32)
    with torch.no_grad():
        k = torch.from_numpy(keep)[None]
        _, v_last, _, (h_end, c_end) = po.lstm_policy_forward(p, torch.from_numpy(obs)[None], h, c, keep=k)
    tr.collect(forced_act=torch.from_numpy(acts).to(DEV), noise=torch.from_numpy(noise).to(DEV))
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    assert np.array_equal(b["obs"], np.stack(want["obs"], 1))
    assert np.array_equal(b["done"], np.stack(want["done"], 1))
    assert np.array_equal(b["keep"], np.stack(want["keep"], 1))
    assert np.array_equal(b["act"], acts)
    assert np.allclose(b["rew"], np.stack(want["rew"], 1), atol=1e-6, rtol=0)
    assert np.allclose(b["val"], np.stack(want["val"], 1), atol=2e-5, rtol=1e-4)
    assert np.allclose(b["logp"], np.stack(want["logp"], 1), atol=2e-5, rtol=1e-4)
    assert np.array_equal(tr.cur_obs.cpu().numpy(), obs)
    assert np.allclose(tr.last_val.cpu().numpy(), v_last[0].numpy(), atol=2e-5, rtol=1e-4)
    # recurrent state handed to the next rollout: masked where the last step ended an episode
    km = torch.from_numpy(keep)[:, None]
    assert np.allclose(tr.h[0].cpu().numpy(), (h[0] * km).numpy(), atol=2e-5)
    assert np.allclose(tr.c[0].cpu().numpy(), (c[0] * km).numpy(), atol=2e-5)
    assert b["done"].sum() >= 2 and tr.nan_count.item() == 0


def test_rollout_sampling_statistics_and_determinism():
    """Counter-RNG sampling inside the fused kernel: action frequencies follow the policy's
    probabilities; same (seed, iteration) -> identical rollout, next iteration differs."""
    from uavppo.trainer import VecPPOTrainer
    N, T = 2048, 32
    a = VecPPOTrainer(N, T, "lstm", hidden=64, device=DEV, seed=9, use_curriculum=False)
    b = VecPPOTrainer(N, T, "lstm", hidden=64, device=DEV, seed=9, use_curriculum=False)
    a.collect()
    b.collect()
    for k in a.buf:
        assert torch.equal(a.buf[k], b.buf[k]), k
    first = a.buf["act"].clone()
    a.iteration += 1
    a.collect()
    assert not torch.equal(first, a.buf["act"])
    # actor init gain 0.01 -> nearly uniform policy; logp must equal log p(a) of a uniform-ish policy
    freq = torch.bincount(first.reshape(-1).long(), minlength=5).float() / first.numel()
    assert (freq - 0.2).abs().max() < 0.01
    assert (a.buf["logp"].exp().mean() - 0.2).abs() < 0.01
    assert torch.isfinite(a.buf["rew"]).all() and a.nan_count.item() == 0


# --------------------------------------------------------------------------------------------- update
def oracle_lstm_update(p, adam, obs, act, rew, val, logp, done, keep, h0, c0, epochs, gae_mode, last_val, num_minibatches=1,
                       grads_out=None):
    """_update_model semantics (train_ppo2.0.py:15-88) with the LSTM policy, torch-CPU autograd.
    num_minibatches > 1: train_ppo2.0.py:43-53's minibatch loop with minibatch m = the whole sequences of envs
    [m*N/M, (m+1)*N/M) (an LSTM minibatch cannot cut a sequence), one optimiser step per minibatch.
    grads_out: a list that receives every optimiser step's UNclipped gradient as a {name: tensor} dict."""
    if gae_mode == "reference_exact":
        adv = po.gae_reference_exact(rew, val, done)
    else:
        adv = po.gae_standard(rew, val, done, last_val)
    adv, ret = po.normalise(adv, val)
    N, T = rew.shape
    x = torch.from_numpy(obs).transpose(0, 1)
    k = torch.from_numpy(keep).transpose(0, 1)
    log = []
    M = num_minibatches
    nb = N // M
    adv2, ret2 = adv.reshape(N, T), ret.reshape(N, T)
    for _ in range(epochs):
        for m in range(M):
            sl = slice(m * nb, (m + 1) * nb)
            leaf = {n: v.detach().clone().requires_grad_(True) for n, v in p.items()}
            probs, value, _, _ = po.lstm_policy_forward(leaf, x[:, sl], h0[:, sl], c0[:, sl], keep=k[:, sl])
            probs = probs.transpose(0, 1).reshape(nb * T, -1)
            value = value.transpose(0, 1).reshape(-1)
            total, pl, vl, ent = po.ppo_losses(probs, value, torch.from_numpy(act[sl]).reshape(-1),
                                               torch.from_numpy(logp[sl]).reshape(-1), adv2[sl].reshape(-1),
                                               ret2[sl].reshape(-1), torch.from_numpy(val[sl]).reshape(-1))
            total.backward()
            grads = {n: leaf[n].grad for n in p}
            if grads_out is not None:
                grads_out.append({n: g.detach().clone() for n, g in grads.items()})
            gn = po.clip_grads(grads)
            adam.step(p, grads)
            log.append([float(pl), float(vl), float(ent), gn])
    return np.array(log), adv.numpy(), ret.numpy()


def _fill_synthetic(tr, N, T, L, H, seed):
    rng = np.random.RandomState(seed)
    d = {"obs": rng.rand(N, T, 6).astype(np.float32), "act": rng.randint(0, 5, (N, T)).astype(np.int32),
         "rew": rng.randn(N, T).astype(np.float32), "val": rng.randn(N, T).astype(np.float32),
         "logp": (np.log(0.2) + 0.1 * rng.randn(N, T)).astype(np.float32),
         "done": (rng.rand(N, T) < 0.08).astype(np.float32)}
    d["keep"] = np.ones((N, T), np.float32)
    d["keep"][:, 1:] = 1 - d["done"][:, :-1]
    last_val = rng.randn(N).astype(np.float32)
    for k, v in d.items():
        if k in tr.buf:
            tr.buf[k].copy_(torch.from_numpy(v))
    if tr.last_val is not None:
        tr.last_val.copy_(torch.from_numpy(last_val))
    return d, last_val


@pytest.mark.parametrize("H,N,T,M,mode", [(64, 8, 24, 2, "reference_exact"), (64, 12, 9, 4, "standard"),
                                          (128, 20, 33, 4, "reference_exact"), (128, 6, 40, 2, "standard"),
                                          (128, 48, 16, 3, "reference_exact")])
def test_lstm_minibatched_update_matches_oracle(H, N, T, M, mode):
    """U1 at M > 1 (train_ppo2.0.py:43-53): minibatches of whole env sequences -- h0/c0 slices, nb-sized work buffers,
    no rollout-forward reuse, loss mean over the minibatch -- against the oracle update that slices the same way."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, device=DEV, seed=3, gae_mode=mode, use_curriculum=False, epochs=2,
                       num_minibatches=M)
    d, last_val = _fill_synthetic(tr, N, T, 1, H, seed=N + M)
    h0 = torch.randn(1, N, H) * 0.3
    c0 = torch.randn(1, N, H) * 0.3
    tr.h0.copy_(h0)
    tr.c0.copy_(c0)
    p = cpu_params(tr.policy)
    adam = po.AdamState(p)
    tr.record = True
    tr.update()
    log, adv, ret = oracle_lstm_update(p, adam, d["obs"], d["act"], d["rew"], d["val"], d["logp"], d["done"], d["keep"],
                                       h0, c0, 2, mode, last_val, num_minibatches=M)
    assert len(tr.log) == 2 * M
    assert np.allclose(tr.adv_n.cpu().numpy().reshape(-1), adv, atol=2e-5, rtol=1e-4)
    n = (N // M) * T
    for i, (sums, gn) in enumerate(tr.log):
        s = sums.cpu().numpy()
        assert np.allclose(s[:3] / n, log[i, :3], rtol=2e-4, atol=2e-6), (i, s[:3] / n, log[i])
        assert np.isclose(gn.item(), log[i, 3], rtol=2e-3), (i, gn.item(), log[i, 3])
    got = cpu_params(tr.policy)
    for k in p:
        assert torch.allclose(got[k], p[k], atol=4e-6, rtol=0), (k, (got[k] - p[k]).abs().max().item())
    assert np.allclose(tr.losses(), log[-1, :3], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("N,T,M,mode", [(8, 32, 2, "reference_exact"), (12, 20, 4, "standard")])
def test_mlp_minibatched_update_matches_oracle(N, T, M, mode):
    """U1 at M > 1 for the reference's MLP policy: minibatch m = the rows of envs [m*N/M, (m+1)*N/M)."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(N, T, "mlp", device=DEV, seed=5, gae_mode=mode, use_curriculum=False, epochs=2, num_minibatches=M)
    d, last_val = _fill_synthetic(tr, N, T, 0, 0, seed=N * M)
    p = cpu_params(tr.policy)
    adam = po.AdamState(p)
    tr.record = True
    tr.update()
    adv = (po.gae_reference_exact(d["rew"], d["val"], d["done"]) if mode == "reference_exact"
           else po.gae_standard(d["rew"], d["val"], d["done"], last_val))
    adv, ret = po.normalise(adv, d["val"])
    adv2, ret2 = adv.reshape(N, T), ret.reshape(N, T)
    nb = N // M
    log = []
    for _ in range(2):
        for m in range(M):
            sl = slice(m * nb, (m + 1) * nb)
            leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
            probs, value, _ = po.mlp_forward(leaf, torch.from_numpy(d["obs"][sl]).reshape(nb * T, 6))
            total, pl, vl, ent = po.ppo_losses(probs, value, torch.from_numpy(d["act"][sl]).reshape(-1),
                                               torch.from_numpy(d["logp"][sl]).reshape(-1), adv2[sl].reshape(-1),
                                               ret2[sl].reshape(-1), torch.from_numpy(d["val"][sl]).reshape(-1))
            total.backward()
            grads = {k: leaf[k].grad for k in p}
            gn = po.clip_grads(grads)
            adam.step(p, grads)
            log.append([float(pl), float(vl), float(ent), gn])
    assert len(tr.log) == 2 * M
    for i, (sums, gn) in enumerate(tr.log):
        s = sums.cpu().numpy()
        assert np.allclose(s[:3] / (nb * T), log[i][:3], rtol=2e-4, atol=2e-6), (i, s[:3] / (nb * T), log[i])
        assert np.isclose(gn.item(), log[i][3], rtol=2e-3), (i, gn.item(), log[i][3])
    got = cpu_params(tr.policy)
    for k in p:
        assert torch.allclose(got[k], p[k], atol=3e-6, rtol=0), (k, (got[k] - p[k]).abs().max().item())


@pytest.mark.parametrize("H,L,N,T,mode", [(64, 1, 6, 24, "reference_exact"), (128, 1, 18, 40, "standard"),
                                          (128, 2, 9, 17, "reference_exact"), (256, 2, 7, 11, "standard")])
def test_lstm_ppo_update_matches_oracle(H, L, N, T, mode):
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, layers=L, device=DEV, seed=3, gae_mode=mode, use_curriculum=False, epochs=3)
    rng = np.random.RandomState(N)
    obs = rng.rand(N, T, 6).astype(np.float32)
    act = rng.randint(0, 5, (N, T)).astype(np.int32)
    rew = rng.randn(N, T).astype(np.float32)
    val = rng.randn(N, T).astype(np.float32)
    logp = (np.log(0.2) + 0.1 * rng.randn(N, T)).astype(np.float32)
    done = (rng.rand(N, T) < 0.08).astype(np.float32)
    keep = np.ones((N, T), np.float32)
    keep[:, 1:] = 1 - done[:, :-1]
    last_val = rng.randn(N).astype(np.float32)
    h0 = torch.randn(L, N, H) * 0.3
    c0 = torch.randn(L, N, H) * 0.3
    for k, v in (("obs", obs), ("act", act), ("rew", rew), ("val", val), ("logp", logp), ("done", done), ("keep", keep)):
        tr.buf[k].copy_(torch.from_numpy(v))
    tr.h0.copy_(h0)
    tr.c0.copy_(c0)
    if tr.last_val is not None:
        tr.last_val.copy_(torch.from_numpy(last_val))
    p = cpu_params(tr.policy)
    adam = po.AdamState(p)
    tr.record = True
    tr.update()
    log, adv, ret = oracle_lstm_update(p, adam, obs, act, rew, val, logp, done, keep, h0, c0, 3, mode, last_val)
    assert np.allclose(tr.adv_n.cpu().numpy().reshape(-1), adv, atol=2e-5, rtol=1e-4)
    n = N * T
    for i, (sums, gn) in enumerate(tr.log):
        s = sums.cpu().numpy()
        assert np.allclose(s[:3] / n, log[i, :3], rtol=2e-4, atol=2e-6), (i, s[:3] / n, log[i])
        assert np.isclose(gn.item(), log[i, 3], rtol=2e-3), (i, gn.item(), log[i, 3])
    got = cpu_params(tr.policy)
    for k in p:
        assert torch.allclose(got[k], p[k], atol=3e-6, rtol=0), (k, (got[k] - p[k]).abs().max().item())


def test_mlp_update_matches_reference_golden(golden):
    """_update_model of the reference (golden L256 case) through the trainer's HIP update path."""
    from uavppo.trainer import VecPPOTrainer
    g = golden("policy_update.npz")
    for case in ("L256", "L7", "L7b", "L1"):
        L = len(g[f"{case}/rew"])
        tr = VecPPOTrainer(1, L, "mlp", device=DEV, seed=0, use_curriculum=False)
        tr.policy.load_state_dict({k: g["init/" + k] for k in po.MLP_KEYS})
        for k, name in (("obs", "obs"), ("rew", "rew"), ("val", "val"), ("logp", "logp"), ("done", "done")):
            tr.buf[k].copy_(torch.from_numpy(g[f"{case}/{name}"]).reshape(tr.buf[k].shape))
        tr.buf["act"].copy_(torch.from_numpy(g[f"{case}/act"].astype(np.int32)).reshape(1, L))
        tr.record = True
        tr.update()
        total = [(s[0] + s[1] - 0.01 * s[2]).item() / L for s, _ in tr.log]
        assert np.allclose(total, g[f"{case}/loss"], rtol=1e-5, atol=2e-6), (case, total, g[f"{case}/loss"])
        assert np.allclose([gn.item() for _, gn in tr.log], g[f"{case}/gnorm"], rtol=2e-4), case
        sd = tr.policy.state_dict()
        for k in po.MLP_KEYS:
            if f"{case}/post/{k}" in g:
                assert np.allclose(sd[k].cpu().numpy(), g[f"{case}/post/{k}"], rtol=0, atol=3e-7), (case, k)
            moved = np.abs(sd[k].cpu().double().numpy() - g["init/" + k]).sum()
            assert np.isclose(moved, g[f"{case}/post_abs/{k}"], rtol=5e-3), (case, k)


def test_end_to_end_reference_loss_curve(golden):
    """BASELINE north_star: 'PPO loss curve matching reference to 1e-4'.  N=1, T=256, MLP policy,
    the reference's recorded actions; env noise/fields regenerated from the recorded numpy seed."""
    from uavppo.trainer import VecPPOTrainer
    g = golden("e2e_v20.npz")
    act = g["act"].astype(np.int32)
    steps = len(act)
    ora = OracleEnv("v2.0", seed=int(g["env_seed"]))
    ora.reset()                                     # MethaneEnv() + the loop's first reset
    episodes = [(ora.source.copy(), ora.conc, ora.tke)]
    noise = np.zeros((steps, 2))
    for t in range(steps):
        st = ora.rs.get_state()
        noise[t] = ora.rs.randn(2)
        ora.rs.set_state(st)
        _, _, d, _, _ = ora.step(int(act[t]))
        if d:
            ora.reset()
            episodes.append((ora.source.copy(), ora.conc, ora.tke))
    bank = FieldBank(np.stack([e[0] for e in episodes]), np.stack([e[1] for e in episodes]),
                     np.stack([e[2] for e in episodes]))
    tr = VecPPOTrainer(1, 256, "mlp", variant="v2.0", device=DEV, bank=bank.interleaved(), bank_sources=bank.sources)
    tr.policy.load_state_dict({k: g["init/" + k] for k in po.MLP_KEYS})
    tr.record = True
    losses, gnorms = [], []
    for it in range(steps // 256):
        sl = slice(it * 256, (it + 1) * 256)
        tr.collect(forced_act=torch.from_numpy(act[sl][None]).to(DEV), noise=torch.from_numpy(noise[sl][None]).to(DEV))
        assert np.array_equal(tr.buf["obs"].cpu().numpy()[0], g["obs"][sl]), it
        assert np.allclose(tr.buf["rew"].cpu().numpy()[0], g["rew"][sl].astype(np.float32), atol=1e-6), it
        assert np.array_equal(tr.buf["done"].cpu().numpy()[0] > 0, g["done"][sl]), it
        # parameters drift by f32 re-association through 120 Adam steps; 1e-4 is the north-star tolerance
        assert np.allclose(tr.buf["val"].cpu().numpy()[0], g["val"][sl], atol=1e-4), it
        assert np.allclose(tr.buf["logp"].cpu().numpy()[0], g["logp"][sl], atol=1e-4), it
        tr.log.clear()
        tr.update()
        tr.update_curriculum()
        tr.iteration += 1
        losses += [(s[0] + s[1] - 0.01 * s[2]).item() / 256 for s, _ in tr.log]
        gnorms += [gn.item() for _, gn in tr.log]
    assert len(losses) == 120
    print("max |loss - reference| over 120 optimiser steps:", np.max(np.abs(np.array(losses) - g["loss"])))
    assert np.max(np.abs(np.array(losses) - g["loss"])) < 1e-4
    assert np.allclose(gnorms, g["gnorm"], rtol=5e-3)
    sd = tr.policy.state_dict()
    for k in po.MLP_KEYS:
        # a parameter with a near-zero gradient can differ by ~one Adam step (lr = 3e-5) after 120 steps
        assert np.isclose(sd[k].double().sum().item(), g["post_sum/" + k], rtol=1e-4, atol=1e-4), k
    assert tr.curriculum.current_radius == 50.0 and len(tr.curriculum.success_history) == len(g["curriculum"])


def test_stepwise_lstm_rollout_matches_fused_kernel():
    """The step-wise rollout (stacked / wide policies) and the fused persistent kernel are two schedules of
    the same computation: identical buffers for a single-layer h=128 policy on the same injected inputs."""
    from uavppo.trainer import VecPPOTrainer
    N, T, H = 21, 30, 128
    bank = FieldBank.from_seed(2 * N, "v2.0", seed=41)
    mk = lambda: VecPPOTrainer(N, T, "lstm", hidden=H, device=DEV, seed=8, bank=bank.interleaved(),
                               bank_sources=bank.sources, gae_mode="standard", use_curriculum=False)
    a, b = mk(), mk()
    a.radius = b.radius = 60.0
    a.reset(); b.reset()
    rng = np.random.RandomState(1)
    fa = torch.from_numpy(rng.randint(1, 5, (N, T)).astype(np.int32)).to(DEV)
    nz = torch.from_numpy(rng.randn(N, T, 2)).to(DEV)
    a.collect(forced_act=fa, noise=nz)
    b.h0.copy_(b.h); b.c0.copy_(b.c)
    b._collect_stepwise_lstm(fa, nz)
    for k in ("obs", "act", "done", "keep", "flags"):
        assert torch.equal(a.buf[k], b.buf[k]), k
    for k in ("rew", "val", "logp"):
        assert torch.allclose(a.buf[k], b.buf[k], atol=2e-5, rtol=1e-4), k
    assert torch.allclose(a.h, b.h, atol=2e-5) and torch.allclose(a.c, b.c, atol=2e-5)
    assert torch.allclose(a.last_val, b.last_val, atol=2e-5)


def test_c5_shaped_policy_trains():
    """BASELINE config C5 family (stacked x2, h=256) end to end at a small size: step-wise rollout, generic
    LSTM forward/backward, fused loss; finite losses and parameters that move."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(48, 16, "lstm", hidden=256, layers=2, device=DEV, seed=2, use_curriculum=False, epochs=2)
    p0 = tr.policy.flat.clone()
    tr.train_iteration()
    pl, vl, ent = tr.losses()
    assert np.isfinite([pl, vl, ent]).all() and 1.5 < ent < 1.61
    assert torch.isfinite(tr.policy.flat).all() and (tr.policy.flat - p0).abs().max() > 1e-6


def test_epoch0_reuses_rollout_forward():
    """The fused rollout kernel's stash/y == what lstm_fwd recomputes with the same parameters, and an
    update that adopts it lands on the same parameters as one that recomputes the forward pass."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    mk = lambda reuse: VecPPOTrainer(40, 24, "lstm", hidden=128, device=DEV, seed=4, use_curriculum=False, epochs=2)
    a, b = mk(True), mk(False)
    b.reuse_rollout_forward = False
    a.collect(); b.collect()
    assert torch.equal(a.buf["obs"], b.buf["obs"]) and a._rollout_forward_valid and not b._rollout_forward_valid
    v = a.policy.views
    y, hn, cn, stash = ops.lstm_fwd(a.buf["obs"], a.buf["keep"], a.h0[0], a.c0[0], v["lstm.weight_ih_l0"],
                                    v["lstm.weight_hh_l0"], v["lstm.bias_ih_l0"], v["lstm.bias_hh_l0"])
    H = 128
    assert torch.allclose(a.work["y0"], y, atol=2e-6)
    assert torch.allclose(a.work["stash0"][..., :5 * H], stash[..., :5 * H], atol=2e-6)
    a.update(); b.update()
    # the two forward passes differ only in f32 accumulation order (~1e-7); Adam turns a gradient g into a step of
    # lr * g / (|g| + eps), which amplifies that on elements whose gradient is tiny: bound the difference by a tenth
    # of the largest possible movement (2 epochs x lr = 6e-5) and require it to be negligible on average
    diff = (a.policy.flat - b.policy.flat).abs()
    assert diff.max().item() < 0.1 * 2 * 3e-5 and diff.mean().item() < 1e-7
    assert torch.allclose(a.loss_sums, b.loss_sums, rtol=1e-5)


@pytest.mark.parametrize("N,T,H", [(1, 3, 64), (1, 256, 128), (17, 1, 128), (33, 130, 64)])
def test_lstm_trainer_edge_shapes(N, T, H):
    """Single env, single step, ragged tiles and horizons that are not multiples of the staging chunks."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, device=DEV, seed=N + T, epochs=2)
    p0 = tr.policy.flat.clone()
    for _ in range(2):
        tr.train_iteration()
    pl, vl, ent = tr.losses()
    assert np.isfinite([pl, vl, ent]).all() and torch.isfinite(tr.policy.flat).all()
    if N * T > 1:                      # a single sample normalises to zero advantage; the value loss still moves it
        assert (tr.policy.flat - p0).abs().max() > 0
    # GAE of the collected buffer against the oracle
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    assert np.allclose(tr.adv.cpu().numpy(), po.gae_reference_exact(b["rew"], b["val"], b["done"]), rtol=2e-5, atol=2e-5)


def test_stepper_rollout_equals_per_call_rollout_and_is_adopted():
    """h = 256 stacked: the stepper rollout (uav_lstm_stepper_*: weights split once, state on the device, stash / y / heads
    written into the update's arrays) gives the SAME rollout as one uav_lstm_fwd call per layer and step, bit for bit; its
    stash equals what the sequence forward recomputes, and an update that adopts it ends on the same parameters."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    mk = lambda: VecPPOTrainer(72, 10, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=5, trend_k=2, epochs=2)
    a, b = mk(), mk()
    b.use_stepper = False
    for it in range(2):                       # second iteration: state handed over from the first, episodes restarting
        a.collect(); b.collect()
        assert a._rollout_forward_valid and not b._rollout_forward_valid
        for k in ("obs", "act", "rew", "val", "logp", "done", "keep", "flags"):
            assert torch.equal(a.buf[k], b.buf[k]), (it, k)
        assert torch.equal(a.h, b.h) and torch.equal(a.c, b.c)
        x = a.buf["obs"]
        for l in range(2):
            v = a.policy.views
            y, _, _, stash = ops.lstm_fwd(x, a.buf["keep"], a.h0[l], a.c0[l], v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"],
                                          v[f"lstm.bias_ih_l{l}"], v[f"lstm.bias_hh_l{l}"])
            assert torch.equal(a.work[f"y{l}"], y) and torch.equal(a.work[f"stash{l}"][..., :5 * stash.shape[-1] // 6], stash[..., :5 * stash.shape[-1] // 6]), (it, l)
            x = y
        heads = ops.gemm(x.reshape(-1, 256), a.policy.views["head.weight"], trans_b=True, bias=a.policy.views["head.bias"])
        assert torch.allclose(a.work["heads"].reshape(-1, 6), heads, atol=1e-6)
        a.update(); b.update()
        a.update_curriculum(); b.update_curriculum()
        a.iteration += 1; b.iteration += 1
        assert torch.equal(a.policy.flat, b.policy.flat), it


def test_fused_rollout_tail_equals_the_five_launches():
    """uav_rollout_tail (heads + action draw + env step + PPOBuffer.store + next observation of a step as ONE launch) against
    the five launches it replaces (uav_gemm_f32's few-column kernel, uav_policy_sample_at, uav_env_step, uav_store_transition,
    the observation copy): every buffer, the heads, the carried state and the env blobs BIT-identical -- free-running, with
    injected step noise, and with forced actions; episodes end and restart inside the horizon (radius 200)."""
    from uavppo.trainer import VecPPOTrainer
    mk = lambda: VecPPOTrainer(70, 12, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=11, trend_k=2, epochs=1,
                               use_curriculum=False)
    a, b = mk(), mk()
    a.radius = b.radius = 200.0
    assert a.use_fused_tail
    b.use_fused_tail = False
    g = torch.Generator(device="cpu").manual_seed(3)
    for it, (fa, nz) in enumerate(((None, None), (None, torch.randn(70, 12, 2, generator=g, dtype=torch.float64).to(DEV)),
                                   (torch.randint(0, 5, (70, 12), generator=g, dtype=torch.int32).to(DEV), None), (None, None))):
        a.collect(forced_act=fa, noise=nz); b.collect(forced_act=fa, noise=nz)
        for k in ("obs", "act", "rew", "val", "logp", "done", "keep", "flags"):
            assert torch.equal(a.buf[k], b.buf[k]), (it, k)
        assert torch.equal(a.work["heads"], b.work["heads"]), it
        assert torch.equal(a.cur_obs, b.cur_obs) and torch.equal(a.h, b.h) and torch.equal(a.c, b.c), it
        assert torch.equal(a.env_state, b.env_state), it
        assert a.buf["done"].sum() > 0 or it == 0
        a.iteration += 1; b.iteration += 1
    assert int(a.nan_count.item()) == int(b.nan_count.item()) == 0


def test_rollout_tail_refuses_bad_arguments():
    """uav_rollout_tail's argument checks answer with the reference's error convention (RuntimeError carrying the text of
    uav_last_error) before anything is launched: a time step outside the horizon, a head count it has no kernel for, a hidden
    size beyond one 256-wide slab."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(8, 4, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=1, trend_k=2, epochs=1, use_curriculum=False)
    tr.collect()
    b, st, v = tr.buf, tr._st, tr.policy.views
    args = lambda t, heads, w, bh, y: (tr.env_state, tr.env_cfg(), y, t, w, bh, heads, st["act"], tr.cur_obs, b["obs"], st["keep"], b["act"],
                                       b["val"], b["logp"], b["keep"], b["rew"], b["done"], b["flags"], tr.nan_count)
    y = tr.work["y1"]
    before = {k: x.clone() for k, x in b.items()}
    with pytest.raises(RuntimeError, match="t=4"):
        ops.rollout_tail(*args(4, tr.work["heads"], v["head.weight"], v["head.bias"], y))
    with pytest.raises(RuntimeError, match="n_act=7"):
        ops.rollout_tail(*args(0, torch.zeros(8, 4, 8, device=DEV), torch.zeros(8, 256, device=DEV), torch.zeros(8, device=DEV), y))
    with pytest.raises(RuntimeError, match="hidden 320"):
        ops.rollout_tail(*args(0, tr.work["heads"], torch.zeros(6, 320, device=DEV), v["head.bias"], torch.zeros(8, 4, 320, device=DEV)))
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], b[k]) for k in before)


def test_update_forward_on_the_steppers_equals_layer_by_layer_calls():
    """The update's forward passes of a stacked h = 256 policy run on the rollout's steppers (all layers step by step, the
    layer above reading the piece planes of the layer below) instead of one uav_lstm_fwd per layer: same kernels, so after two
    iterations (2 x 3 optimiser steps, two of them on recomputed forwards each) gradient and parameters are BIT-identical."""
    from uavppo.trainer import VecPPOTrainer
    mk = lambda: VecPPOTrainer(72, 11, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=9, trend_k=2, epochs=3)
    a, b = mk(), mk()
    b.policy.use_stepper_forward = False
    for it in range(2):
        a.train_iteration(); b.train_iteration()
        assert torch.equal(a.policy.grad, b.policy.grad), it
        assert torch.equal(a.policy.flat, b.policy.flat), it
        for l in range(2):
            assert torch.equal(a.work[f"y{l}"], b.work[f"y{l}"]) and torch.equal(a.work[f"stash{l}"][..., :5 * a.work[f"stash{l}"].shape[-1] // 6], b.work[f"stash{l}"][..., :5 * a.work[f"stash{l}"].shape[-1] // 6]), (it, l)


def test_pipelined_stack_backward_equals_layer_by_layer():
    """uav_lstm_bwd_stack (the layers' BPTTs pipelined on internal streams, the layer below one step behind the one above)
    runs the kernels of one uav_lstm_bwd per layer: after two iterations (4 optimiser steps) the parameters are BIT-identical."""
    from uavppo.trainer import VecPPOTrainer
    mk = lambda: VecPPOTrainer(80, 12, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=7, trend_k=2, epochs=2)
    a, b = mk(), mk()
    b.policy.use_stack_bwd = False
    for it in range(2):
        a.train_iteration(); b.train_iteration()
        assert torch.equal(a.policy.grad, b.policy.grad), it
        assert torch.equal(a.policy.flat, b.policy.flat), it
    assert "dgates0" in a.work and "dgates0" not in b.work


def test_c5_trend_policy_trains():
    """C5 shape family with the trend channels: obs_dim 8, stacked h=256 LSTM, step-wise rollout."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(32, 12, "lstm", hidden=256, layers=2, variant="v2.1", device=DEV, seed=3, trend_k=2,
                       use_curriculum=False, epochs=2)
    assert tr.buf["obs"].shape == (32, 12, 8) and tr.policy.views["lstm.weight_ih_l0"].shape == (1024, 8)
    tr.train_iteration()
    o = tr.buf["obs"]
    assert torch.allclose(o[:, 1:, 6], o[:, 1:, 2] - o[:, :-1, 2], atol=1e-6) or bool((tr.buf["done"] > 0).any())
    assert np.isfinite(tr.losses()).all()
    # a fused-kernel-sized policy with trend obs also works (falls back to the step-wise rollout)
    tr2 = VecPPOTrainer(40, 10, "lstm", hidden=128, device=DEV, seed=3, trend_k=1, use_curriculum=False, epochs=1)
    tr2.train_iteration()
    assert np.isfinite(tr2.losses()).all() and tr2.buf["obs"].shape[-1] == 7


# --------------------------------------------------------------------------------------------- range guard
def _oracle_one_update(tr, d, h0, c0, last_val, mode, epochs):
    p = cpu_params(tr.policy)
    adam = po.AdamState(p)
    log, _, _ = oracle_lstm_update(p, adam, d["obs"], d["act"], d["rew"], d["val"], d["logp"], d["done"], d["keep"],
                                   h0, c0, epochs, mode, last_val)
    return p, log


@pytest.mark.parametrize("what", ["weights", "obs", "h0"])
def test_range_guard_switches_to_wide_range_kernels(what):
    """include/uavppo.h range note: the fp16-split kernels need |w| < 65504, |x| < 4096, |h0| < 64.  The trainer measures
    all three (uav_absmax / uav_clip_adam's pmax_out) and runs the bf16-split kernels when one is violated -- without
    an environment variable -- and counts the event; results still match the f32 oracle."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    N, T, H = 10, 12, 64
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, device=DEV, seed=3, use_curriculum=False, epochs=1, lr=1e-6)
    d, last_val = _fill_synthetic(tr, N, T, 1, H, seed=5)
    h0 = torch.randn(1, N, H) * 0.3
    c0 = torch.randn(1, N, H) * 0.3
    if what == "weights":
        with torch.no_grad():       # one recurrent weight far outside fp16's range (the unit saturates; everything stays finite)
            tr.policy.views["lstm.weight_hh_l0"][5, 7] = 1.0e5
    elif what == "obs":
        d["obs"][..., 0] *= 1.0e4
        tr.buf["obs"].copy_(torch.from_numpy(d["obs"]))
    else:
        h0[0, :, 3] = 100.0
    tr.h0.copy_(h0)
    tr.c0.copy_(c0)
    want, log = _oracle_one_update(tr, d, h0, c0, last_val, "reference_exact", 1)
    tr.record = True
    tr.update()
    assert tr.arith == "bf16x6" and tr.range_events == 1 and ops.get_lstm_arith(DEV) == "bf16x6"
    s = tr.log[0][0].cpu().numpy()
    assert np.isfinite(s).all()
    assert np.allclose(s[:3] / (N * T), log[0, :3], rtol=3e-4, atol=3e-6), (s[:3] / (N * T), log[0])
    assert np.isclose(tr.log[0][1].item(), log[0, 3], rtol=3e-3)
    # in range again -> back to the fp16 split on the next probe
    tr2 = VecPPOTrainer(N, T, "lstm", hidden=H, device=DEV, seed=3, use_curriculum=False, epochs=1)
    _fill_synthetic(tr2, N, T, 1, H, seed=5)
    tr2.update()
    assert tr2.arith == "fp16x3" and tr2.range_events == 0 and ops.get_lstm_arith(DEV) == "fp16x3"


def test_range_guard_in_the_training_loop():
    """Weights pushed out of fp16's range between iterations: the next rollout takes the step-wise path on the bf16-split
    kernels (uav_rollout has only the fp16 form), training stays finite; in range it stays on the fused kernel."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(64, 16, "lstm", hidden=128, device=DEV, seed=4, epochs=2)
    tr.train_iteration()
    assert tr.arith == "fp16x3" and tr.range_events == 0
    with torch.no_grad():
        tr.policy.views["lstm.weight_hh_l0"][0, 0] = 7.0e4
    tr.train_iteration()
    assert tr.arith == "bf16x6" and tr.range_events >= 1
    assert np.isfinite(tr.losses()).all() and torch.isfinite(tr.policy.flat).all()
    with torch.no_grad():
        tr.policy.views["lstm.weight_hh_l0"][0, 0] = 0.01
    tr.train_iteration()
    tr.train_iteration()
    assert tr.arith == "fp16x3"


def test_range_guard_covers_h256():
    """BASELINE C5's policy (h = 256 x 2, trend channels): a weight outside fp16's range switches the trainer to the wide
    mode, where the h = 256 kernels run the generic exact-f32 step path (no stepper, no fp16 pieces); training stays
    finite and the guard returns to the fp16 split once the weight is back in range."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(32, 6, "lstm", hidden=256, layers=2, device=DEV, seed=4, epochs=1, trend_k=2)
    tr.train_iteration()
    assert tr.arith == "fp16x3" and tr.range_events == 0
    with torch.no_grad():
        tr.policy.views["lstm.weight_hh_l1"][3, 3] = 9.0e4
    tr.train_iteration()
    assert tr.arith == "bf16x6" and tr.range_events >= 1
    assert np.isfinite(tr.losses()).all() and torch.isfinite(tr.policy.flat).all()
    with torch.no_grad():
        tr.policy.views["lstm.weight_hh_l1"][3, 3] = 0.01
    tr.train_iteration()
    tr.train_iteration()
    assert tr.arith == "fp16x3"


def test_range_guard_covers_the_fused_mlp():
    """The fused MLP kernels run the 256 x 128 layer on the fp16 split (max |param| < 2048).  A LayerNorm gain of 3000
    pushes the layer's input past fp16's range: the trainer must switch the handle to the exact-f32 form of the same
    kernels by itself, and the update must still match the f32 oracle; in range it stays on the split."""
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    N, T = 12, 20
    for big in (True, False):
        tr = VecPPOTrainer(N, T, "mlp", device=DEV, seed=5, use_curriculum=False, epochs=1)
        if big:
            with torch.no_grad():
                tr.policy.views["feature.1.weight"][7] = 3000.0
        d, last_val = _fill_synthetic(tr, N, T, 0, 0, seed=3)
        p = cpu_params(tr.policy)
        adam = po.AdamState(p)
        tr.record = True
        tr.update()
        assert (tr.arith != "fp16x3") == big and (tr.range_events == 1) == big
        assert (ops.get_lstm_arith(DEV) != "fp16x3") == big
        adv = po.gae_reference_exact(d["rew"], d["val"], d["done"])
        adv, ret = po.normalise(adv, d["val"])
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
        probs, value, _ = po.mlp_forward(leaf, torch.from_numpy(d["obs"]).reshape(N * T, 6))
        total, pl, vl, ent = po.ppo_losses(probs, value, torch.from_numpy(d["act"]).reshape(-1), torch.from_numpy(d["logp"]).reshape(-1),
                                           adv, ret, torch.from_numpy(d["val"]).reshape(-1))
        total.backward()
        gn = po.clip_grads({k: leaf[k].grad for k in p})
        s = tr.log[0][0].cpu().numpy()
        assert np.isfinite(s).all()
        assert np.allclose(s[:3] / (N * T), [float(pl), float(vl), float(ent)], rtol=3e-4, atol=3e-6)
        assert np.isclose(tr.log[0][1].item(), gn, rtol=3e-3)
    ops.set_lstm_arith("fp16x3")


def test_adam_publishes_max_abs_param(ops=None):
    from uavppo import ops
    n = 5000
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g).to(DEV)
    p[1234] = -77.0
    grad = (torch.randn(n, generator=g) * 1e-3).to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    out = torch.full((1,), -1.0, device=DEV)
    ops.clip_adam(p, grad, m, v, 1, 1e-3, pmax_out=out)
    assert out.item() == p.abs().max().item() and 76.9 < out.item() < 77.1
    x = torch.randn(100000, generator=g).to(DEV)
    assert ops.absmax(x).item() == x.abs().max().item()
    x[77] = float("nan")
    assert ops.absmax(x).item() == float("inf")


# --------------------------------------------------------------------------------------------- fused MLP kernels
@pytest.mark.parametrize("n", [1, 7, 16, 100, 256, 5000, 70001])
def test_fused_mlp_gradient_matches_oracle_and_layered_path(n):
    """uav_mlp_ppo_grad (forward + loss + backward on chip, csrc/mlp_fused.hip) vs torch-CPU autograd through the oracle's
    restatement of model.py:42-53 + train_ppo2.0.py:55-83, and vs the layer-by-layer HIP path; ragged last tiles included."""
    from uavppo import ops
    from uavppo.policy import MLPActorCritic
    pol = MLPActorCritic(6, 5, device=DEV, seed=n)
    with torch.no_grad():        # non-trivial LayerNorm parameters and biases
        g = torch.Generator().manual_seed(1)
        for k in ("feature.0.bias", "feature.1.bias", "feature.3.bias", "feature.4.bias", "head.bias"):
            pol.views[k].copy_(torch.randn(pol.views[k].shape, generator=g) * 0.2)
        for k in ("feature.1.weight", "feature.4.weight"):
            pol.views[k].copy_(1 + 0.3 * torch.randn(pol.views[k].shape, generator=g))
        pol.views["head.weight"].mul_(20.0)
    rng = np.random.RandomState(n)
    obs = rng.rand(n, 6).astype(np.float32)
    act = rng.randint(0, 5, n).astype(np.int32)
    adv = rng.randn(n).astype(np.float32)
    ret = rng.randn(n).astype(np.float32)
    vo = rng.randn(n).astype(np.float32)
    lp = (np.log(0.2) + 0.3 * rng.randn(n)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    sums = torch.zeros(4, dtype=torch.float64, device=DEV)
    ops.mlp_ppo_grad(pol.flat, d(obs), d(act), d(lp), d(adv), d(ret), d(vo), 1.0 / n, 0.2, 0.01, sums, pol.grad)
    got = {k: v.detach().cpu().clone() for k, v in pol.named_grads().items()}
    got_sums = sums.cpu().numpy()
    # oracle
    leaf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in pol.named_views().items()}
    probs, value, _ = po.mlp_forward(leaf, torch.from_numpy(obs))
    total, pl, vl, ent = po.ppo_losses(probs, value, torch.from_numpy(act), torch.from_numpy(lp), torch.from_numpy(adv),
                                       torch.from_numpy(ret), torch.from_numpy(vo))
    total.backward()
    assert np.allclose(got_sums[:3] / n, [float(pl), float(vl), float(ent)], rtol=2e-5, atol=1e-6) and got_sums[3] == 0
    for k in leaf:
        scale = leaf[k].grad.abs().max().item() + 1e-12
        err = (got[k] - leaf[k].grad).abs().max().item()
        # f32 sums over n samples on both sides, in different orders: rounding grows like sqrt(n) * 2^-24
        assert err <= (2e-5 + 2e-7 * np.sqrt(n)) * scale + 1e-9, (k, err, scale)
    # layer-by-layer HIP path
    heads = pol.heads(d(obs))
    dheads = torch.empty(n, 6, device=DEV)
    s2 = torch.zeros(4, dtype=torch.float64, device=DEV)
    ops.ppo_loss_heads(heads, d(act), d(lp), d(adv), d(ret), d(vo), 1.0 / n, 0.2, 0.01, s2, dheads)
    g2 = pol.backward(dheads).clone()
    ops.mlp_ppo_grad(pol.flat, d(obs), d(act), d(lp), d(adv), d(ret), d(vo), 1.0 / n, 0.2, 0.01, sums, pol.grad)
    assert torch.allclose(pol.grad, g2, rtol=2e-4, atol=2e-6 * g2.abs().max().item())
    assert np.allclose(sums.cpu().numpy(), s2.cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("pattern", ["growing", "shrinking", "spiky"])
def test_fused_mlp_split_gradient_over_a_wide_dynamic_range(pattern):
    """The fp16-split products of uav_mlp_ppo_grad against an f64 autograd of the same network, next to the exact-f32 form
    of the same kernel (uav_set_lstm_arith): sample gradients spanning 30 decades -- growing along the batch (every tile
    lowers the dW2 wave scale and rescales the accumulators), shrinking (late tiles far below the running scale), and a
    few huge samples among tiny ones (per-sample dz2 scale of the da1 product).  The split kernels must be as close to f64
    as the exact-f32 ones (to a factor 2 + f32 noise), tensor by tensor."""
    from uavppo import ops
    from uavppo.policy import MLPActorCritic
    n = 4096
    pol = MLPActorCritic(6, 5, device=DEV, seed=11)
    with torch.no_grad():
        g = torch.Generator().manual_seed(2)
        for k in ("feature.1.weight", "feature.4.weight"):
            pol.views[k].copy_(1 + 0.3 * torch.randn(pol.views[k].shape, generator=g))
        pol.views["head.weight"].mul_(20.0)
    rng = np.random.RandomState(5)
    obs = rng.rand(n, 6).astype(np.float32)
    act = rng.randint(0, 5, n).astype(np.int32)
    lp = (np.log(0.2) + 0.3 * rng.randn(n)).astype(np.float32)
    vo = rng.randn(n).astype(np.float32)
    expo = {"growing": np.linspace(-20, 10, n), "shrinking": np.linspace(10, -20, n),
            "spiky": np.where(rng.rand(n) < 0.01, 8.0, -12.0)}[pattern]
    mag = (10.0 ** expo).astype(np.float32)
    adv = (rng.randn(n) * mag).astype(np.float32)
    # value-loss gradients of the same spread on the small side; capped at 1e2, because for |ret| >> |V| the two branches of
    # max((V - R)^2, (Vclip - R)^2) TIE in f32 and not in f64 -- a property of the f32 loss, not of the kernels
    ret = (vo + rng.randn(n) * np.minimum(mag, 1e2)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    leaf = {k: v.detach().cpu().double().requires_grad_(True) for k, v in pol.named_views().items()}
    probs, value, _ = po.mlp_forward(leaf, torch.from_numpy(obs).double())
    total, _, _, _ = po.ppo_losses(probs, value, torch.from_numpy(act), torch.from_numpy(lp).double(), torch.from_numpy(adv).double(),
                                   torch.from_numpy(ret).double(), torch.from_numpy(vo).double())
    total.backward()
    errs = {}
    for mode in ("fp16x3", "f32_mfma"):
        with ops.lstm_arith(mode):
            sums = torch.zeros(4, dtype=torch.float64, device=DEV)
            ops.mlp_ppo_grad(pol.flat, d(obs), d(act), d(lp), d(adv), d(ret), d(vo), 1.0 / n, 0.2, 0.01, sums, pol.grad)
            got = {k: v.detach().cpu().double().clone() for k, v in pol.named_grads().items()}
        assert all(torch.isfinite(v).all() for v in got.values()), mode
        errs[mode] = {k: ((got[k] - leaf[k].grad).abs().max() / (leaf[k].grad.abs().max() + 1e-300)).item() for k in leaf}
    for k in leaf:
        assert errs["fp16x3"][k] < 2e-5, (pattern, k, errs["fp16x3"][k])
        assert errs["fp16x3"][k] <= 2.0 * errs["f32_mfma"][k] + 3e-7, (pattern, k, errs["fp16x3"][k], errs["f32_mfma"][k])


def test_fused_mlp_split_kernels_at_the_edge_of_their_operand_range():
    """include/uavppo.h: the fp16-split form of uav_mlp_ppo_grad needs max |param| < 2048 (a1 <= sqrt(255) |g1| + |be1| < 65504).
    With a LayerNorm-1 gain of 2000 and a weight of W2 at 2000 -- just inside -- it must still agree with the exact-f32 form
    of the same kernel (the trainer's guard switches at half the limit; the kernel's own limit is what is checked here)."""
    from uavppo import ops
    from uavppo.policy import MLPActorCritic
    n = 2048
    pol = MLPActorCritic(6, 5, device=DEV, seed=4)
    with torch.no_grad():
        pol.views["feature.1.weight"][11] = 2000.0
        pol.views["feature.1.weight"][200] = -1500.0
        pol.views["feature.3.weight"][5, 9] = 2000.0
    rng = np.random.RandomState(9)
    d = lambda a: torch.from_numpy(a).to(DEV)
    obs, act = rng.rand(n, 6).astype(np.float32), rng.randint(0, 5, n).astype(np.int32)
    adv, ret, vo = (rng.randn(n).astype(np.float32) for _ in range(3))
    lp = (np.log(0.2) + 0.3 * rng.randn(n)).astype(np.float32)
    out = {}
    for mode in ("fp16x3", "f32_mfma"):
        with ops.lstm_arith(mode):
            sums = torch.zeros(4, dtype=torch.float64, device=DEV)
            ops.mlp_ppo_grad(pol.flat, d(obs), d(act), d(lp), d(adv), d(ret), d(vo), 1.0 / n, 0.2, 0.01, sums, pol.grad)
            out[mode] = (pol.grad.clone(), sums.clone())
    g3, gf = out["fp16x3"][0].double(), out["f32_mfma"][0].double()
    assert torch.isfinite(g3).all() and torch.isfinite(gf).all()
    assert ((g3 - gf).norm() / gf.norm()).item() < 2e-5
    assert torch.allclose(out["fp16x3"][1], out["f32_mfma"][1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("N,T", [(5, 40), (37, 70)])
def test_fused_mlp_rollout_matches_oracle_simulation(N, T):
    """uav_rollout policy_kind 0 (the reference's policy, train_ppo2.0.py:157-198 for N envs) vs a step-by-step oracle
    simulation on injected noise + forced actions + a materialised bank: env bit-exact, policy outputs to f32 tolerance;
    and the step-wise HIP path (uav_mlp_fwd + uav_policy_sample + uav_env_step) fills identical buffers."""
    from uavppo.trainer import VecPPOTrainer
    bank = FieldBank.from_seed(3 * N, "v2.0", seed=31)
    mk = lambda: VecPPOTrainer(N, T, "mlp", variant="v2.0", device=DEV, seed=5, bank=bank.interleaved(),
                               bank_sources=bank.sources, gae_mode="standard", use_curriculum=False, log_info=True)
    tr, ts = mk(), mk()
    assert tr.fused_mlp
    ts.fused_mlp = False
    for t_ in (tr, ts):
        t_.radius = 45.0
        t_.reset()
    rng = np.random.RandomState(2)
    noise = rng.randn(N, T, 2)
    ora = OracleVecEnv(N, bank, "v2.0", radius=45.0)
    obs = ora.reset()
    p = cpu_params(tr.policy)
    acts = np.zeros((N, T), np.int32)
    want = {k: [] for k in ("obs", "rew", "done", "val", "logp")}
    for t in range(T):
        a = []
        for i, e in enumerate(ora.envs):
            dd = e.source - e.pos
            hom = (3 if dd[0] > 0 else 4) if abs(dd[0]) > abs(dd[1]) else (1 if dd[1] > 0 else 2)
            a.append(hom if (t < 30 or i % 2 == 0) else int(rng.randint(0, 5)))
        acts[:, t] = a
        with torch.no_grad():
            probs, value, _ = po.mlp_forward(p, torch.from_numpy(obs))
            lp = po.categorical_logp(probs, torch.tensor(a))
        want["obs"].append(obs.copy())
        want["val"].append(value[:, 0].numpy().copy())
        want["logp"].append(lp.numpy().copy())
        obs, rew, done, reached, info, term = ora.step(np.array(a), noise[:, t])
        want["rew"].append(rew.astype(np.float32))
        want["done"].append(done.astype(np.float32))
    with torch.no_grad():
        _, v_last, _ = po.mlp_forward(p, torch.from_numpy(obs))
    fa, nz = torch.from_numpy(acts).to(DEV), torch.from_numpy(noise).to(DEV)
    tr.collect(forced_act=fa, noise=nz)
    ts.collect(forced_act=fa, noise=nz)
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    assert np.array_equal(b["obs"], np.stack(want["obs"], 1))
    assert np.array_equal(b["done"], np.stack(want["done"], 1))
    assert np.array_equal(b["act"], acts)
    assert np.allclose(b["rew"], np.stack(want["rew"], 1), atol=1e-6, rtol=0)
    assert np.allclose(b["val"], np.stack(want["val"], 1), atol=2e-5, rtol=1e-4)
    assert np.allclose(b["logp"], np.stack(want["logp"], 1), atol=2e-5, rtol=1e-4)
    assert np.array_equal(tr.cur_obs.cpu().numpy(), obs)
    assert np.allclose(tr.last_val.cpu().numpy(), v_last[:, 0].numpy(), atol=2e-5, rtol=1e-4)
    assert b["done"].sum() >= 2 and tr.nan_count.item() == 0
    for k in ("obs", "act", "done", "flags"):
        assert torch.equal(tr.buf[k], ts.buf[k]), k
    for k in ("rew", "val", "logp"):
        assert torch.allclose(tr.buf[k], ts.buf[k], atol=2e-5, rtol=1e-4), k
    assert torch.allclose(tr.info, ts.info, atol=1e-4) and torch.allclose(tr.last_val, ts.last_val, atol=2e-5)


def test_fused_mlp_rollout_logp_is_the_updates_first_forward():
    """Rollout and update run the same forward code in the same order: at epoch 0 the ratio is exactly 1, so the policy loss
    is exactly -mean(adv_n) and sampling follows the policy (counter RNG, statistics)."""
    from uavppo.trainer import VecPPOTrainer
    N, T = 1024, 32
    tr = VecPPOTrainer(N, T, "mlp", device=DEV, seed=9, use_curriculum=False, epochs=1)
    tr.collect()
    tr.record = True
    tr.update()
    s = tr.log[0][0].cpu().numpy()
    assert abs(s[0] / (N * T) + tr.adv_n.double().mean().item()) < 1e-9
    freq = torch.bincount(tr.buf["act"].reshape(-1).long(), minlength=5).float() / (N * T)
    assert (freq - 0.2).abs().max() < 0.02 and (tr.buf["logp"].exp().mean() - 0.2).abs() < 0.01


@pytest.mark.parametrize("N,T", [(1, 3), (17, 1), (1, 256)])
def test_fused_mlp_edge_shapes(N, T):
    """Single env, single step, ragged 16-env tile: fused MLP rollout + update stay finite and move the parameters;
    GAE of the collected buffer against the oracle."""
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(N, T, "mlp", device=DEV, seed=N + T, epochs=2)
    assert tr.fused_mlp
    p0 = tr.policy.flat.clone()
    for _ in range(2):
        tr.train_iteration()
    pl, vl, ent = tr.losses()
    assert np.isfinite([pl, vl, ent]).all() and torch.isfinite(tr.policy.flat).all()
    if N * T > 1:
        assert (tr.policy.flat - p0).abs().max() > 0
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    assert np.allclose(tr.adv.cpu().numpy(), po.gae_reference_exact(b["rew"], b["val"], b["done"]), rtol=2e-5, atol=2e-5)


def test_new_entry_points_reject_bad_arguments():
    """Error convention of the ABI (non-zero status -> RuntimeError with uav_last_error's text) on the round-2 entry points."""
    from uavppo import ops
    from uavppo.policy import MLPActorCritic
    pol = MLPActorCritic(8, 5, device=DEV, seed=0)            # 8 inputs: not the reference's network
    n = 32
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)
    with pytest.raises(RuntimeError, match="fused kernel is the reference's network"):
        ops.mlp_ppo_grad(pol.flat, z(n, 8), z(n, dt=torch.int32), z(n), z(n), z(n), z(n), 1.0 / n, 0.2, 0.01,
                         z(4, dt=torch.float64), pol.grad, in_dim=8)
    with pytest.raises(KeyError):
        ops.set_lstm_arith("fp8", DEV)
    from uavppo import _lib
    assert _lib.lib().uav_set_lstm_arith(ops.Context.get(DEV).handle, 7) != 0
    assert b"uav_set_lstm_arith" in _lib.lib().uav_last_error()
    with pytest.raises(RuntimeError, match="expected a contiguous tensor|expected shape"):
        ops.absmax(z(8, 8)[:, ::2])


def test_device_curriculum_equals_host_curriculum_in_the_loop():
    """The curriculum on the device (default: no host synchronisation in train_iteration) against the host-side class fed by the
    same success bits: after every iteration the same episode counts and window length, radius / bonus to 1e-12 (device pow vs
    libm), and -- the env kernels reading them from the device block -- the same rollouts bit for bit."""
    from uavppo.trainer import VecPPOTrainer
    kw = dict(policy="lstm", hidden=64, device=DEV, seed=5, epochs=1)
    a = VecPPOTrainer(512, 64, device_curriculum=True, **kw)
    b = VecPPOTrainer(512, 64, device_curriculum=False, **kw)
    for tr in (a, b):
        tr.radius = 200.0            # many episodes end: the 120-episode window fills several times per rollout
        tr.reset()
    b.curriculum.current_radius = 200.0      # (host mode: the class keeps its own copy; the device block IS the state)
    assert a.device_curriculum and not b.device_curriculum
    for it in range(5):
        for tr in (a, b):
            tr.train_iteration()
        assert a.episodes_done == b.episodes_done and a.successes_done == b.successes_done, it
        assert len(a.curriculum.success_history) == len(b.curriculum.success_history)
        assert np.isclose(a.radius, b.radius, rtol=1e-12, atol=0) and np.isclose(float(a.bonus), float(b.bonus), rtol=1e-12, atol=0)
        assert isinstance(a.bonus, np.float64) == isinstance(b.bonus, np.float64)
        if a.radius == b.radius and float(a.bonus) == float(b.bonus):
            assert torch.equal(a.buf["obs"], b.buf["obs"]) and torch.equal(a.buf["flags"], b.buf["flags"])
    assert a.episodes_done > 600 and a.radius < 200.0
    # the lagged mirror the training script reads (never waits): at most one rollout behind
    assert 0 < a.episodes_lagged <= a.episodes_done
