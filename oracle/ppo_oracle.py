"""CPU oracle of the PPO update path (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates (torch-CPU fp32 tensors / numpy f32 scalars, own structure):
  M2  PPOActorCritic.forward    PPOV2.0/model.py:42-53
  G1  GAE scan                  PPOV2.0/train_ppo2.0.py:18-32   (+ textbook form PPOV1.0/ppo0.0.py:337-350)
  G2  normalise + returns       PPOV2.0/train_ppo2.0.py:35-40
  U2  clipped-PPO loss          PPOV2.0/train_ppo2.0.py:55-83   (Categorical(probs) semantics of torch)
  U3  clip_grad_norm_ + Adam    PPOV2.0/train_ppo2.0.py:85-88,114
  T1  PPOTrainer.update         PPOV2.0/model.py:131-164
  L1  nn.LSTM semantics         PPOV2.0/model.py:206-212, PPOV2.1/model.py:263 (i,f,g,o; b_ih+b_hh)

Parity pin: tests/golden/{policy,update,curriculum,e2e}_*.npz, all produced by the
reference itself (oracle/gen_golden.py); the LSTM part is pinned against torch.nn.LSTM.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

GAMMA, LAMBDA = 0.99, 0.95                 # config.py:12-13
CLIP_EPSILON, ENTROPY_BETA = 0.2, 0.01     # config.py:14-15
LEARNING_RATE, EPOCHS = 3e-5, 5            # config.py:16,18
MAX_GRAD_NORM = 0.5                        # train_ppo2.0.py:87
F32_EPS = float(np.finfo(np.float32).eps)

MLP_KEYS = ("feature.0.weight", "feature.0.bias", "feature.1.weight", "feature.1.bias",
            "feature.3.weight", "feature.3.bias", "feature.4.weight", "feature.4.bias",
            "actor.weight", "actor.bias", "critic.weight", "critic.bias")


# ----------------------------------------------------------------------------- policy (M2)
def mlp_forward(p, x):
    """p: dict name->tensor with the reference's state_dict keys. Returns probs, value[B,1], logits."""
    h = F.linear(x, p["feature.0.weight"], p["feature.0.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[-1],), p["feature.1.weight"], p["feature.1.bias"], 1e-5))
    h = F.linear(h, p["feature.3.weight"], p["feature.3.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[-1],), p["feature.4.weight"], p["feature.4.bias"], 1e-5))
    logits = F.linear(h, p["actor.weight"], p["actor.bias"])
    if torch.isnan(logits).any():
        raise RuntimeError("NaN in model output")             # model.py:47-49
    value = F.linear(h, p["critic.weight"], p["critic.bias"])
    return torch.softmax(logits, dim=-1), value, logits


def categorical_logp(probs, actions):
    """log_prob of torch.distributions.Categorical(probs): renormalise, clamp to
    [eps, 1-eps] (probs_to_logits), log, gather.  Reached from train_ppo2.0.py:64-65,189."""
    q = probs / probs.sum(-1, keepdim=True)
    q = q.clamp(min=F32_EPS, max=1 - F32_EPS)
    return torch.log(q).gather(-1, actions.long().unsqueeze(-1)).squeeze(-1)


# ----------------------------------------------------------------------------- GAE (G1, G2)
def gae_reference_exact(rew, val, done, gamma=GAMMA, lam=LAMBDA):
    """train_ppo2.0.py:18-32 in f32 scalar arithmetic, same operation order.

    Quirks kept: the mask of step t comes from done[t+1]; the last step bootstraps from its
    own value; nothing resets at episode boundaries other than through that mask.
    Accepts [T] or [N,T] (independent rows).
    """
    rew = np.asarray(rew, np.float32)
    val = np.asarray(val, np.float32)
    done = np.asarray(done, np.float32)
    if rew.ndim == 2:
        return np.stack([gae_reference_exact(rew[i], val[i], done[i], gamma, lam)
                         for i in range(rew.shape[0])])
    T = rew.shape[0]
    adv = np.zeros(T, np.float32)
    g = np.float32(gamma)
    gl = np.float32(gamma * lam)
    one = np.float32(1.0)
    last = np.float32(0.0)
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            nnt = one - done[t]
            nv = val[t] * nnt
        else:
            nnt = one - done[t + 1]
            nv = val[t + 1] * nnt
        delta = (rew[t] + g * nv) - val[t]
        last = delta + (gl * nnt) * last
        adv[t] = last
    return adv


def gae_standard(rew, val, done, last_val, gamma=GAMMA, lam=LAMBDA):
    """Textbook GAE (mask done[t], bootstrap V(s_T)); returns advantages [N,T] or [T]."""
    rew = np.asarray(rew, np.float32)
    val = np.asarray(val, np.float32)
    done = np.asarray(done, np.float32)
    if rew.ndim == 2:
        return np.stack([gae_standard(rew[i], val[i], done[i], np.asarray(last_val)[i], gamma, lam)
                         for i in range(rew.shape[0])])
    T = rew.shape[0]
    adv = np.zeros(T, np.float32)
    g, gl, one = np.float32(gamma), np.float32(gamma * lam), np.float32(1.0)
    last = np.float32(0.0)
    for t in range(T - 1, -1, -1):
        nnt = one - done[t]
        nv = (np.float32(last_val) if t == T - 1 else val[t + 1]) * nnt
        delta = (rew[t] + g * nv) - val[t]
        last = delta + (gl * nnt) * last
        adv[t] = last
    return adv


def normalise(adv, val):
    """train_ppo2.0.py:35-40: centre, unbiased std with guard, returns = normalised adv + values."""
    adv = torch.as_tensor(adv, dtype=torch.float32).reshape(-1)
    val = torch.as_tensor(val, dtype=torch.float32).reshape(-1)
    adv = adv - adv.mean()
    std = adv.std()
    if std < 1e-6 or torch.isnan(std):
        std = 1.0
    adv = adv / (std + 1e-6)
    return adv, adv + val


# ----------------------------------------------------------------------------- loss (U2)
def ppo_losses(probs, value, actions, logp_old, adv, ret, val_old,
               clip=CLIP_EPSILON, beta=ENTROPY_BETA):
    """Returns (total, policy_loss, value_loss, entropy) -- train_ppo2.0.py:64-83."""
    if torch.isnan(probs).any():
        raise RuntimeError("NaN in probs")                    # train_ppo2.0.py:58-62
    logp = categorical_logp(probs, actions)
    ratio = (logp - logp_old).exp()
    s1 = ratio * adv
    s2 = torch.clamp(ratio, 1 - clip, 1 + clip) * adv
    pl = -torch.min(s1, s2).mean()
    v = value.reshape(-1)
    vclip = val_old + (v - val_old).clamp(-clip, clip)
    vl = 0.5 * torch.max((v - ret).pow(2), (vclip - ret).pow(2)).mean()
    ent = -torch.sum(probs * torch.log(probs + 1e-8), dim=-1).mean()
    return pl + vl - beta * ent, pl, vl, ent


# ----------------------------------------------------------------------------- optimiser (U3)
class AdamState:
    """torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay=0) restated over a dict."""

    def __init__(self, params, lr=LEARNING_RATE, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k in params:
            g = grads[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def clip_grads(grads, max_norm=MAX_GRAD_NORM):
    """torch.nn.utils.clip_grad_norm_: global L2 norm; scale by max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(total)


def update_model(params, adam, states, actions, rewards, values, logp_old, dones,
                 forward=mlp_forward, epochs=EPOCHS, gae="reference_exact", last_val=None):
    """_update_model (train_ppo2.0.py:15-88) over ONE flat buffer, one full-batch minibatch per
    epoch (the reference's `randperm(L).split(256)` with L == 256: a permutation of a
    mean-reduced batch).  `params` is updated in place.  Returns per-epoch
    [policy_loss, value_loss, entropy, total, grad_norm] and (adv, returns)."""
    if gae == "reference_exact":
        adv = gae_reference_exact(rewards, values, dones)
    else:
        adv = gae_standard(rewards, values, dones, last_val)
    adv, ret = normalise(adv, values)
    x = torch.as_tensor(states, dtype=torch.float32)
    a = torch.as_tensor(actions).long()
    lp = torch.as_tensor(logp_old, dtype=torch.float32)
    vo = torch.as_tensor(values, dtype=torch.float32)
    log = []
    for _ in range(epochs):
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        probs, value, _ = forward(leaf, x)
        total, pl, vl, ent = ppo_losses(probs, value, a, lp, adv, ret, vo)
        total.backward()
        grads = {k: leaf[k].grad for k in params}
        gn = clip_grads(grads)
        adam.step(params, grads)
        log.append([float(pl), float(vl), float(ent), float(total), gn])
    return np.asarray(log, np.float64), adv.numpy(), ret.numpy()


# ----------------------------------------------------------------------------- curriculum (T1)
class CurriculumOracle:
    """PPOTrainer.update (model.py:131-164); `env_radius`/`env_bonus` are what the env sees
    (pushed at the START of each call, so they lag the trainer by one episode)."""

    INITIAL_RADIUS, MIN_RADIUS, RADIUS_DECAY = 50.0, 5.0, 0.9   # config.py:27-29
    SUCCESS_THRESHOLD, WINDOW = 0.6, 120                        # config.py:30-31
    EXPLORE_BONUS, DECAY_FACTOR = 0.6, 0.999                    # config.py:21-22

    def __init__(self):
        self.radius = self.INITIAL_RADIUS
        self.bonus = self.EXPLORE_BONUS
        self.env_radius = self.INITIAL_RADIUS
        self.env_bonus = self.EXPLORE_BONUS
        self.history = []

    def update(self, success):
        self.env_radius, self.env_bonus = self.radius, self.bonus
        self.history.append(bool(success))
        if len(self.history) > self.WINDOW:
            self.history.pop(0)
        full = len(self.history) >= self.WINDOW
        if full:
            rate = sum(self.history) / len(self.history)
            self.bonus = self.bonus * (self.DECAY_FACTOR ** (1 + rate))
        self.bonus = max(self.bonus, 0.1)
        if full:
            if rate > self.SUCCESS_THRESHOLD:
                self.radius = max(self.MIN_RADIUS, self.radius *
                                  (self.RADIUS_DECAY ** (2 + 3 * (rate - self.SUCCESS_THRESHOLD))))
            elif rate < 0.25:
                self.radius = min(self.INITIAL_RADIUS, self.radius * 1.1)
            if abs(self.radius - self.env_radius) > 5:
                self.radius = self.env_radius + 5 * math.copysign(1.0, self.radius - self.env_radius)
            self.history = []


# ----------------------------------------------------------------------------- LSTM (L1)
def lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, keep=None):
    """One nn.LSTM layer, time-major x[T,N,I].  keep[T,N] (1/0) multiplies the incoming
    (h,c) of step t (0 = the episode ended at t-1: the recurrent state restarts from 0)."""
    T = x.shape[0]
    h, c = h0, c0
    ys = []
    for t in range(T):
        if keep is not None:
            k = keep[t].unsqueeze(-1)
            h, c = h * k, c * k
        g = F.linear(x[t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys), h, c


def lstm_policy_forward(p, x, h0, c0, keep=None):
    """LSTM actor-critic of BASELINE.json: nn.LSTM(obs, H, L) -> actor Linear(H,A), critic
    Linear(H,1).  p uses nn.LSTM's parameter names under 'lstm.' plus 'actor.*','critic.*'.
    x[T,N,I]; h0,c0 [L,N,H].  Returns probs[T,N,A], value[T,N], logits, (hn, cn)."""
    L = h0.shape[0]
    y = x
    hn, cn = [], []
    for l in range(L):
        y, h, c = lstm_layer_forward(y, h0[l], c0[l], p[f"lstm.weight_ih_l{l}"],
                                     p[f"lstm.weight_hh_l{l}"], p[f"lstm.bias_ih_l{l}"],
                                     p[f"lstm.bias_hh_l{l}"], keep)
        hn.append(h)
        cn.append(c)
    logits = F.linear(y, p["actor.weight"], p["actor.bias"])
    value = F.linear(y, p["critic.weight"], p["critic.bias"]).squeeze(-1)
    return torch.softmax(logits, -1), value, logits, (torch.stack(hn), torch.stack(cn))


def init_lstm_policy(obs_dim, hidden, layers, n_act=5, seed=0):
    """Random-init parameters (nn.LSTM default U(-1/sqrt(H),1/sqrt(H)); heads as the MLP's)."""
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(hidden)
    p = {}
    for l in range(layers):
        i = obs_dim if l == 0 else hidden
        p[f"lstm.weight_ih_l{l}"] = (torch.rand(4 * hidden, i, generator=g) * 2 - 1) * k
        p[f"lstm.weight_hh_l{l}"] = (torch.rand(4 * hidden, hidden, generator=g) * 2 - 1) * k
        p[f"lstm.bias_ih_l{l}"] = (torch.rand(4 * hidden, generator=g) * 2 - 1) * k
        p[f"lstm.bias_hh_l{l}"] = (torch.rand(4 * hidden, generator=g) * 2 - 1) * k
    p["actor.weight"] = (torch.rand(n_act, hidden, generator=g) * 2 - 1) * k * 0.1
    p["actor.bias"] = torch.zeros(n_act)
    p["critic.weight"] = (torch.rand(1, hidden, generator=g) * 2 - 1) * k
    p["critic.bias"] = torch.zeros(1)
    return p
