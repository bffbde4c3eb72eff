"""Oracle (CPU restatement) vs golden vectors produced by the reference's model/_update_model/PPOTrainer."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle.env_oracle import OracleEnv


def _init(g, prefix="init/"):
    return {k: torch.from_numpy(g[prefix + k].copy()) for k in po.MLP_KEYS}


def test_policy_forward(golden):
    g = golden("policy_update.npz")
    p = _init(g)
    probs, value, _ = po.mlp_forward(p, torch.from_numpy(g["fwd_x"]))
    assert np.allclose(probs.numpy(), g["fwd_probs"], rtol=0, atol=1e-7)
    assert np.allclose(value.numpy(), g["fwd_value"], rtol=0, atol=1e-6)
    assert sum(v.numel() for v in p.values()) == 36230


@pytest.mark.parametrize("case", ["L256", "L7", "L7b", "L1"])
def test_update_model(golden, case):
    g = golden("policy_update.npz")
    p = _init(g)
    adam = po.AdamState(p)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        log, adv, ret = po.update_model(p, adam, g[f"{case}/obs"], g[f"{case}/act"], g[f"{case}/rew"],
                                        g[f"{case}/val"], g[f"{case}/logp"], g[f"{case}/done"])
    assert np.allclose(log[:, 3], g[f"{case}/loss"], rtol=1e-6, atol=1e-6)
    assert np.allclose(log[:, 4], g[f"{case}/gnorm"], rtol=1e-5)
    for k in po.MLP_KEYS:
        if f"{case}/post/{k}" in g:
            assert np.allclose(p[k].numpy(), g[f"{case}/post/{k}"], rtol=0, atol=2e-7), k
        moved = np.abs(p[k].double().numpy() - g["init/" + k]).sum()
        assert np.isclose(moved, g[f"{case}/post_abs/{k}"], rtol=2e-3), k
        if f"{case}/m/{k}" in g:
            assert np.allclose(adam.m[k].numpy(), g[f"{case}/m/{k}"], rtol=1e-4, atol=1e-7), k
            assert np.allclose(adam.v[k].numpy(), g[f"{case}/v/{k}"], rtol=1e-4, atol=1e-10), k


def test_gae_quirks():
    """done[t+1] masks step t; last step bootstraps from itself; L=1 -> std NaN guard."""
    r = np.array([1, 2, 3, 4], np.float32)
    v = np.array([.5, .25, .125, 1], np.float32)
    d = np.array([0, 0, 1, 0], np.float32)
    a = po.gae_reference_exact(r, v, d)
    g, gl = np.float32(.99), np.float32(.99 * .95)
    a3 = (r[3] + g * v[3]) - v[3]
    a2 = (r[2] + g * v[3]) - v[2] + gl * a3
    a1 = r[1] - v[1]                       # masked by done[2]
    a0 = (r[0] + g * v[1]) - v[0] + gl * a1
    assert np.allclose(a, [a0, a1, a2, a3], rtol=1e-6)
    adv, ret = po.normalise(np.array([3.0], np.float32), np.array([1.0], np.float32))
    assert float(adv[0]) == 0.0 and float(ret[0]) == 1.0
    s = po.gae_standard(r, v, d, 2.0)
    assert np.isclose(s[3], r[3] + g * 2.0 - v[3]) and np.isclose(s[2], r[2] - v[2])


def test_curriculum(golden):
    g = golden("curriculum.npz")
    names = sorted({k.split("/")[0] for k in g.files if "/" in k})
    assert len(names) == 5
    for n in names:
        c = po.CurriculumOracle()
        for i, s in enumerate(g[f"{n}/seq"]):
            c.update(bool(s))
            got = [c.radius, c.bonus, c.env_radius, c.env_bonus]
            assert np.allclose(got, g[f"{n}/trace"][i], rtol=1e-14, atol=0), (n, i)


def test_end_to_end_loss_curve(golden):
    """N=1 replay of 24 reference updates with recorded actions: values/logp per step and
    the loss curve (BASELINE north_star: 'PPO loss curve matching reference to 1e-4')."""
    g = golden("e2e_v20.npz")
    p = _init(g)
    adam = po.AdamState(p)
    env = OracleEnv("v2.0", seed=int(g["env_seed"]))
    cur = po.CurriculumOracle()
    state = env.reset()      # MethaneEnv() resets in __init__, the loop resets again (train_ppo2.0.py:112,139)
    buf = {k: [] for k in ("s", "a", "r", "v", "lp", "d")}
    losses, gn = [], []
    T = len(g["act"])
    for t in range(T):
        assert np.array_equal(state, g["obs"][t]), t
        with torch.no_grad():
            probs, value, _ = po.mlp_forward(p, torch.from_numpy(state)[None])
            lp = po.categorical_logp(probs, torch.tensor([int(g["act"][t])]))
        assert abs(float(value) - g["val"][t]) < 2e-5 and abs(float(lp) - g["logp"][t]) < 2e-5, t
        o, r, d, s, _ = env.step(int(g["act"][t]))
        assert r == g["rew"][t] and d == g["done"][t]
        for k, x in zip(("s", "a", "r", "v", "lp", "d"), (state, g["act"][t], r, float(value), float(lp), d)):
            buf[k].append(x)
        if len(buf["s"]) >= 256:
            log, _, _ = po.update_model(p, adam, np.stack(buf["s"]), np.array(buf["a"]),
                                        np.array(buf["r"], np.float32), np.array(buf["v"], np.float32),
                                        np.array(buf["lp"], np.float32), np.array(buf["d"], np.float32))
            losses += list(log[:, 3])
            gn += list(log[:, 4])
            buf = {k: [] for k in buf}
        state = o
        if d:
            cur.update(s)
            env.radius, env.bonus = cur.env_radius, cur.env_bonus
            state = env.reset()
    assert len(losses) == len(g["loss"]) == 120
    assert np.max(np.abs(np.array(losses) - g["loss"])) < 1e-4
    assert np.allclose(gn, g["gnorm"], rtol=1e-3)
    for k in po.MLP_KEYS:
        assert np.isclose(p[k].double().sum().item(), g["post_sum/" + k], rtol=1e-4, atol=1e-5), k


def test_lstm_matches_torch_nn_lstm():
    torch.manual_seed(0)
    for (T, N, I, H, L) in [(8, 4, 6, 64, 1), (5, 3, 8, 32, 2)]:
        ref = torch.nn.LSTM(I, H, L)
        p = {"lstm." + k: v.detach() for k, v in ref.named_parameters()}
        p.update({"actor.weight": torch.randn(5, H), "actor.bias": torch.zeros(5),
                  "critic.weight": torch.randn(1, H), "critic.bias": torch.zeros(1)})
        x, h0, c0 = torch.randn(T, N, I), torch.randn(L, N, H), torch.randn(L, N, H)
        y, (hn, cn) = ref(x, (h0, c0))
        probs, value, logits, (h2, c2) = po.lstm_policy_forward(p, x, h0, c0)
        assert torch.allclose(hn, h2, atol=1e-6) and torch.allclose(cn, c2, atol=1e-6)
        assert torch.allclose(logits, y @ p["actor.weight"].T, atol=1e-5)
