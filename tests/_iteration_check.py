"""Shared checker of the whole-iteration parity tests (tests/test_gpu_fullsize.py, tests/test_gpu_fullsize_configs.py):
the trainer's GAE + normalise + EPOCHS optimiser steps (the reference's `_update_model`, PPOV2.0/train_ppo2.0.py:15-88)
on the buffers ITS OWN rollout collected, against the oracle (torch-CPU autograd) on the same buffers.  Three layers:

  (a) every optimiser step's losses and UNclipped gradient against the oracle evaluated AT THE SAME PARAMETERS (the
      trainer's parameters before that step): isolates the kernels' arithmetic -- forward, loss, BPTT, weight gradients --
      at f32 tolerance, step after step;
  (b) every clip + Adam step against the oracle's optimiser driven by the SAME gradients (the trainer's);
  (c) the free-running oracle update (its own parameters, its own Adam state): advantages / returns, the loss curve, the
      gradient norms and the final parameters.  Adam's first step is lr * g / (|g| + eps) = lr * sign(g): an element whose
      gradient cancels to rounding noise steps by +lr on one side and -lr on the other, so the two trajectories differ by up to
      2 lr in a few elements and every LATER gradient by ~1e-3 relative (measured; decaying) -- a property of the optimiser, the
      same between any two f32 implementations; hence (a) and (b), and the Adam-noise tolerances here.

Test infrastructure; imports oracle/ only as the checker."""
import json
import os

import numpy as np
import torch

from oracle import ppo_oracle as po


def cpu_params(policy):
    return {k: v.detach().cpu().clone() for k, v in policy.named_views().items()}


def named_from_flat(policy, flat):
    """A flat gradient / parameter vector cut into the oracle's tensor names (actor / critic rows split out of `head`)."""
    src, o = {}, 0
    for name, shape in policy.layout:
        n = int(np.prod(shape))
        src[name] = flat[o:o + n].view(shape)
        o += n
    return policy._named(src)


def _oracle_step(p, x, k, h0, c0, act, logp, adv, ret, val, sl, nbT):
    """Losses and unclipped gradient of one optimiser step at parameters p (train_ppo2.0.py:55-86)."""
    leaf = {n: v.detach().clone().requires_grad_(True) for n, v in p.items()}
    probs, value, _, _ = po.lstm_policy_forward(leaf, x[:, sl], h0[:, sl], c0[:, sl], keep=k[:, sl])
    probs = probs.transpose(0, 1).reshape(nbT, -1)
    value = value.transpose(0, 1).reshape(-1)
    total, pl, vl, ent = po.ppo_losses(probs, value, act[sl].reshape(-1), logp[sl].reshape(-1), adv[sl].reshape(-1),
                                       ret[sl].reshape(-1), val[sl].reshape(-1))
    total.backward()
    return {n: leaf[n].grad.detach() for n in p}, [float(pl.detach()), float(vl.detach()), float(ent.detach())]


def _rel_l2(got, want):
    num = sum(float(((got[k].double() - want[k].double()) ** 2).sum()) for k in want)
    den = sum(float((want[k].double() ** 2).sum()) for k in want)
    return (num / den) ** 0.5


def update_vs_oracle(tr, label, grad_rel_tol=2e-5, lr_steps_tol=0.1, far_frac=3e-2, loss_rtol=2e-4, loss_atol=2e-6):
    """`tr` has collected a rollout (tr.collect()) and NOT yet updated.  Runs tr.update() with recording, then the three
    comparisons of the module docstring; asserts and returns the measured deviations (also written to
    gpurun_out/parity_<label>.json when that directory exists, for the round's records)."""
    N, T = tr.N, tr.T
    epochs, M = tr.hp["epochs"], tr.num_minibatches
    nb = N // M
    lr = tr.hp["lr"]
    p_free = cpu_params(tr.policy)
    p_start = {k: v.clone() for k, v in p_free.items()}
    h0, c0 = tr.h0.cpu().clone(), tr.c0.cpu().clone()
    b = {k: v.cpu().numpy() for k, v in tr.buf.items()}
    last_val = None if tr.last_val is None else tr.last_val.cpu().numpy()
    tr.record = tr.record_grads = True
    tr.log.clear()
    tr.grad_log.clear()
    tr.update()
    torch.cuda.synchronize()
    assert len(tr.log) == epochs * M == len(tr.grad_log)
    # ---- G1 / G2 on the oracle
    adv = (po.gae_reference_exact(b["rew"], b["val"], b["done"]) if tr.gae_mode == "reference_exact"
           else po.gae_standard(b["rew"], b["val"], b["done"], last_val))
    adv_n, ret = po.normalise(adv, b["val"])
    m = {"label": label, "N": N, "T": T, "samples": N * T, "episode_ends": int(b["done"].sum()), "epochs": epochs, "minibatches": M}
    got_adv, got_ret = tr.adv_n.cpu().numpy().reshape(-1), tr.ret.cpu().numpy().reshape(-1)
    m["adv_max_abs_diff"] = float(np.abs(got_adv - adv_n.numpy()).max())
    m["ret_max_abs_diff"] = float(np.abs(got_ret - ret.numpy()).max())
    assert np.allclose(got_adv, adv_n.numpy(), atol=3e-5, rtol=1e-4), m
    assert np.allclose(got_ret, ret.numpy(), atol=3e-5, rtol=1e-4), m
    x = torch.from_numpy(b["obs"]).transpose(0, 1)
    k = torch.from_numpy(b["keep"]).transpose(0, 1)
    act, logp, val = torch.from_numpy(b["act"]), torch.from_numpy(b["logp"]), torch.from_numpy(b["val"])
    adv2, ret2 = adv_n.reshape(N, T), ret.reshape(N, T)
    adam_free = po.AdamState(p_free)
    p_same = {kk: v.clone() for kk, v in p_start.items()}          # (b): the oracle's optimiser fed the trainer's gradients
    adam_same = po.AdamState(p_same)
    m["steps"] = []
    i = 0
    for _ in range(epochs):
        for mb in range(M):
            sl = slice(mb * nb, (mb + 1) * nb)
            g_hip_flat, p_hip_flat = tr.grad_log[i]
            g_hip = {n: v.clone() for n, v in named_from_flat(tr.policy, g_hip_flat.cpu()).items()}
            p_hip = named_from_flat(tr.policy, p_hip_flat.cpu())
            got_losses = (tr.log[i][0].cpu().numpy()[:3] / (nb * T)).tolist()
            got_gn = float(tr.log[i][1].item())
            # (a) same parameters
            g_at, l_at = _oracle_step(p_hip, x, k, h0, c0, act, logp, adv2, ret2, val, sl, nb * T)
            rel_at = _rel_l2(g_hip, g_at)
            # tensor by tensor: largest deviation over the tensor's largest entry -- or, for a tensor whose gradient cancels
            # to rounding noise (the critic bias at epoch 0 is sum(V - ret) = -sum(adv_n) = 0 by the normalisation itself),
            # over 1e-2 of the whole gradient's largest entry
            gmax = max(float(g.abs().max()) for g in g_at.values())
            per_tensor = {n: float((g_hip[n] - g_at[n]).abs().max()) / max(float(g_at[n].abs().max()), 1e-2 * gmax) for n in g_at}
            gn_at = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g_at.values())))
            # (b) same gradients: the oracle's clip + Adam on the trainer's gradient, from parameters that followed the same rule
            p_same_before = max(float((p_same[n] - p_hip[n]).abs().max()) for n in p_same)
            po.clip_grads(g_hip)
            adam_same.step(p_same, g_hip)
            # (c) free-running oracle
            if i == 0:
                g_free, l_free = g_at, l_at               # same parameters at the first step
            else:
                g_free, l_free = _oracle_step(p_free, x, k, h0, c0, act, logp, adv2, ret2, val, sl, nb * T)
            g_free = {n: v.clone() for n, v in g_free.items()}
            gn_free = po.clip_grads(g_free)
            adam_free.step(p_free, g_free)
            m["steps"].append({"losses": got_losses, "losses_at_same_params": l_at, "losses_free": l_free, "gnorm": got_gn,
                               "gnorm_at_same_params": gn_at, "gnorm_free": gn_free, "grad_rel_l2_at_same_params": rel_at,
                               "grad_max_rel_per_tensor": per_tensor, "adam_param_diff_before_step": p_same_before})
            i += 1
    got_p = cpu_params(tr.policy)
    m["adam_param_max_abs_diff_same_grads"] = max(float((got_p[n] - p_same[n]).abs().max()) for n in p_same)
    n_far = n_all = 0
    worst = 0.0
    for n in p_free:
        d = (got_p[n] - p_free[n]).abs()
        n_far += int((d > lr_steps_tol * lr).sum())
        n_all += d.numel()
        worst = max(worst, float(d.max()))
    moved = max(float((p_free[n] - p_start[n]).abs().max()) for n in p_free)
    m.update({"free_param_max_abs_diff": worst, "free_param_far_count": n_far, "param_count": n_all, "free_param_far_tol": lr_steps_tol * lr,
              "param_max_move_oracle": moved})
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"parity_{label}.json"), "w") as f:
            json.dump(m, f, indent=1)
    print(f"[{label}] whole-iteration parity: " + json.dumps({kk: v for kk, v in m.items() if kk != "steps"}))
    for e, s in enumerate(m["steps"]):
        print(f"[{label}]   step {e}: grad rel L2 at the same parameters {s['grad_rel_l2_at_same_params']:.2e}, gnorm {s['gnorm']:.7g} / same "
              f"{s['gnorm_at_same_params']:.7g} / free {s['gnorm_free']:.7g}; losses {s['losses']} / same {s['losses_at_same_params']} / free {s['losses_free']}")
    for e, s in enumerate(m["steps"]):
        # (a): f32 tolerance.  The policy loss is a mean of O(1) terms that cancels to ~1e-5: absolute tolerance
        assert np.allclose(s["losses"], s["losses_at_same_params"], rtol=2e-6, atol=2e-7), (e, s)
        assert np.isclose(s["gnorm"], s["gnorm_at_same_params"], rtol=2e-5), (e, s)
        assert s["grad_rel_l2_at_same_params"] <= grad_rel_tol, (e, s["grad_rel_l2_at_same_params"])
        assert max(s["grad_max_rel_per_tensor"].values()) <= 50 * grad_rel_tol, (e, s["grad_max_rel_per_tensor"])
        # (c): the loss curve (north_star: 1e-4) and the norms of the free-running oracle
        assert np.allclose(s["losses"], s["losses_free"], rtol=loss_rtol, atol=loss_atol), (e, s)
        assert np.isclose(s["gnorm"], s["gnorm_free"], rtol=2e-3), (e, s)
    # (b): clip + Adam on the same gradients reproduces the trainer's parameters to f32 rounding of an lr-sized step
    assert m["adam_param_max_abs_diff_same_grads"] <= 2e-3 * lr * epochs * M + 1e-9, m
    assert moved >= 0.5 * lr                                   # the update did move the parameters
    assert worst <= 2 * epochs * M * lr, m
    assert n_far <= far_frac * n_all + 2, m
    return m
