"""HIP kernels of the update path vs the oracle, through the C ABI (ctypes).  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x, dtype=None):
    t = torch.as_tensor(np.asarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


@pytest.fixture(scope="module")
def ops():
    from uavppo import ops as o
    return o


def _rollout_like(n, T, seed, p_done=0.02):
    rng = np.random.RandomState(seed)
    rew = (rng.randn(n, T) * 2).astype(np.float32)
    val = rng.randn(n, T).astype(np.float32)
    done = (rng.rand(n, T) < p_done).astype(np.float32)
    return rew, val, done


@pytest.mark.parametrize("n,T", [(1, 256), (1, 1), (3, 7), (5, 64), (4, 65), (9, 128), (2, 300)])
@pytest.mark.parametrize("mode", ["reference_exact", "standard"])
def test_gae_matches_oracle(ops, n, T, mode):
    rew, val, done = _rollout_like(n, T, seed=n * 1000 + T, p_done=0.1)
    done[:, -1] = (np.arange(n) % 2)          # done on the very last step for half the rows
    if T > 2:
        done[0, 0] = 1.0
    last = np.random.RandomState(1).randn(n).astype(np.float32)
    if mode == "reference_exact":
        want = po.gae_reference_exact(rew, val, done)
        got = ops.gae(dev(rew), dev(val), dev(done), 0.99, 0.95, mode)
    else:
        want = po.gae_standard(rew, val, done, last)
        got = ops.gae(dev(rew), dev(val), dev(done), 0.99, 0.95, mode, last_val=dev(last))
    # f32 tolerance: the wave scan re-associates the recurrence (oracle is strictly sequential)
    assert np.allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


def test_gae_golden_case(ops, golden):
    g = golden("policy_update.npz")
    for case in ("L256", "L7", "L7b", "L1"):
        r, v, d = g[f"{case}/rew"][None], g[f"{case}/val"][None], g[f"{case}/done"][None]
        got = ops.gae(dev(r), dev(v), dev(d), 0.99, 0.95).cpu().numpy()
        assert np.allclose(got, po.gae_reference_exact(r, v, d), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(1, 256), (1, 1), (1, 2), (64, 128), (4096, 128)])
def test_normalise_matches_oracle(ops, shape):
    rng = np.random.RandomState(shape[0])
    adv = (rng.randn(*shape) * 3 + 1.5).astype(np.float32)
    val = rng.randn(*shape).astype(np.float32)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_a, want_r = po.normalise(adv, val)
    a = dev(adv)
    stats = ops.adv_stats(a)
    ga, gr = ops.adv_normalise(a, dev(val), stats)
    assert np.allclose(ga.cpu().numpy().reshape(-1), want_a.numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(gr.cpu().numpy().reshape(-1), want_r.numpy(), rtol=1e-5, atol=1e-5)
    s = stats.cpu().numpy()
    assert s[2] == adv.size and np.isclose(s[0], adv.astype(np.float64).sum(), rtol=1e-12, atol=1e-9)


def test_normalise_constant_buffer_guard(ops):
    """std < 1e-6 -> divide by (1 + 1e-6): train_ppo2.0.py:37-39."""
    adv = np.full((2, 8), 0.25, np.float32)
    val = np.arange(16, dtype=np.float32).reshape(2, 8)
    a = dev(adv)
    ga, gr = ops.adv_normalise(a, dev(val), ops.adv_stats(a))
    assert np.allclose(ga.cpu().numpy(), 0.0, atol=1e-7) and np.allclose(gr.cpu().numpy(), val, atol=1e-6)


def _loss_inputs(n, seed, A=5, extreme=False):
    rng = np.random.RandomState(seed)
    logits = (rng.randn(n, A) * (6.0 if extreme else 1.0)).astype(np.float32)
    value = rng.randn(n).astype(np.float32)
    act = rng.randint(0, A, n).astype(np.int32)
    logp_old = (np.log(0.2) + 0.3 * rng.randn(n)).astype(np.float32)
    adv = rng.randn(n).astype(np.float32)
    ret = (value + 0.5 * rng.randn(n)).astype(np.float32)
    val_old = (value + 0.3 * rng.randn(n)).astype(np.float32)
    return logits, value, act, logp_old, adv, ret, val_old


@pytest.mark.parametrize("n,extreme", [(1, False), (7, False), (256, False), (5000, True), (70001, False)])
def test_ppo_loss_fwd_bwd(ops, n, extreme):
    logits, value, act, lpo, adv, ret, vo = _loss_inputs(n, n, extreme=extreme)
    if extreme:
        logits[0] = [60, 0, 0, 0, -60]      # clamp of Categorical(probs) active (q -> 1-eps / eps)
        act[0] = 0
        logits[1] = [60, 0, 0, 0, -60]
        act[1] = 4
    z = torch.tensor(logits, requires_grad=True)
    v = torch.tensor(value, requires_grad=True)
    total, pl, vl, ent = po.ppo_losses(torch.softmax(z, -1), v, torch.tensor(act), torch.tensor(lpo),
                                       torch.tensor(adv), torch.tensor(ret), torch.tensor(vo))
    total.backward()
    sums, dz, dv = ops.ppo_loss(dev(logits), dev(value), dev(act), dev(lpo), dev(adv), dev(ret), dev(vo),
                                1.0 / n, 0.2, 0.01)
    s = sums.cpu().numpy()
    assert s[3] == 0
    assert np.allclose(s[:3] / n, [float(pl), float(vl), float(ent)], rtol=2e-5, atol=1e-6)
    assert np.allclose(dz.cpu().numpy(), z.grad.numpy(), rtol=1e-4, atol=2e-6 / n + 1e-9)
    assert np.allclose(dv.cpu().numpy(), v.grad.numpy(), rtol=1e-4, atol=2e-6 / n + 1e-9)


def test_ppo_loss_nan_is_counted(ops):
    logits, value, act, lpo, adv, ret, vo = _loss_inputs(300, 3)
    logits[17, 2] = np.nan
    sums, _, _ = ops.ppo_loss(dev(logits), dev(value), dev(act), dev(lpo), dev(adv), dev(ret), dev(vo),
                              1.0 / 300, 0.2, 0.01)
    assert sums.cpu().numpy()[3] == 1      # host raises RuntimeError("NaN in probs") on this flag


def test_policy_sample(ops):
    rng = np.random.RandomState(0)
    n = 200000
    logits = np.tile(np.array([[0.5, -1.0, 2.0, 0.0, 1.0]], np.float32), (n, 1))
    p = torch.softmax(torch.tensor(logits[0]), -1).numpy()
    act, logp, probs, nan = ops.policy_sample(dev(logits), seed=7, counter=3, want_probs=True)
    a = act.cpu().numpy()
    freq = np.bincount(a, minlength=5) / n
    assert np.abs(freq - p).max() < 5e-3 and nan.item() == 0
    want_lp = po.categorical_logp(torch.tensor(p)[None].repeat(n, 1), torch.tensor(a)).numpy()
    assert np.allclose(logp.cpu().numpy(), want_lp, atol=1e-6)
    assert np.allclose(probs.cpu().numpy()[0], p, atol=1e-7)
    # injected uniforms = inverse CDF; forced actions pass through
    u = rng.rand(64).astype(np.float32)
    act2, _, _, _ = ops.policy_sample(dev(logits[:64]), u=dev(u))
    cdf = np.cumsum(p / p.sum())
    assert np.array_equal(act2.cpu().numpy(), np.minimum(np.searchsorted(cdf, u, side="right"), 4))
    forced = rng.randint(0, 5, 64).astype(np.int32)
    act3, lp3, _, _ = ops.policy_sample(dev(logits[:64]), forced_act=dev(forced))
    assert np.array_equal(act3.cpu().numpy(), forced)
    # same (seed, counter) -> same draw; different counter -> different draw
    b1 = ops.policy_sample(dev(logits[:4096]), seed=7, counter=3)[0].cpu().numpy()
    b2 = ops.policy_sample(dev(logits[:4096]), seed=7, counter=4)[0].cpu().numpy()
    assert np.array_equal(b1, a[:4096]) and not np.array_equal(b1, b2)


def test_clip_adam_matches_torch(ops):
    torch.manual_seed(0)
    n = 36230
    p0 = torch.randn(n)
    p_ref = {"w": p0.clone()}
    adam = po.AdamState(p_ref)
    p = p0.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    gn = torch.zeros(1, device=DEV)
    for step in range(1, 6):
        g = torch.randn(n) * (3.0 if step % 2 else 1e-3)     # clipped and unclipped steps
        grads = {"w": g.clone()}
        norm = po.clip_grads(grads)
        adam.step(p_ref, grads)
        ops.clip_adam(p, g.to(DEV), m, v, step, 3e-5, gnorm_out=gn)
        assert np.isclose(gn.item(), norm, rtol=1e-5)
        assert torch.allclose(p.cpu(), p_ref["w"], rtol=0, atol=2e-7)
        assert torch.allclose(m.cpu(), adam.m["w"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(v.cpu(), adam.v["w"], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("M,N,K,ta,tb", [(1, 6, 128, False, True), (256, 256, 6, False, True), (300, 128, 256, False, True),
                                         (257, 130, 70, False, False), (6, 128, 5000, True, False),
                                         (512, 134, 40000, True, False), (1000, 6, 128, False, True)])
def test_gemm(ops, M, N, K, ta, tb):
    rng = np.random.RandomState(M + N + K)
    a = rng.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    b = rng.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    got = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias)).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5 * np.sqrt(K))
    c0 = rng.randn(M, N).astype(np.float32)
    out = dev(c0)
    ops.gemm(dev(a), dev(b), ta, tb, out=out, accumulate=True)
    assert np.allclose(out.cpu().numpy(), want - bias + c0, rtol=1e-5, atol=1e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K,ta,tb", [(128, 128, 96, False, False), (256, 256, 1000, False, True), (128, 384, 50, True, False),
                                         (1024, 256, 40000, True, False), (4096, 256, 1024, False, False),
                                         (256, 128, 4100, True, True)])
def test_gemm_split_fp16_has_f32_accuracy(ops, M, N, K, ta, tb):
    """uav_gemm_f16x3 (three fp16 piece products per f32 product) against an f64 product: its error is no larger than the
    exact-f32 MFMA kernel's on the same operands, in all four operand layouts, with and without split-K, K tails included."""
    rng = np.random.RandomState(M + N + K)
    a = rng.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    b = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    f32 = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias)).cpu().numpy()
    got = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), split_fp16=True).cpu().numpy()
    scale = np.abs(want).max()
    e32, e16 = np.abs(f32 - want).max() / scale, np.abs(got - want).max() / scale
    assert e16 <= 1.5 * e32 + 1e-7, (e16, e32)
    c0 = rng.randn(M, N).astype(np.float32)
    out = dev(c0)
    ops.gemm(dev(a), dev(b), ta, tb, out=out, accumulate=True, split_fp16=True)
    assert np.abs(out.cpu().numpy() - (want - bias + c0)).max() / scale <= 1.5 * e32 + 2e-7


@pytest.mark.parametrize("magnitude", [1e-7, 3e-3, 1.0, 2e4, 1e9])
def test_gemm_split_fp16_block_scale(ops, magnitude):
    """A gradient-sized (or huge) A operand: with a_absmax the operand is scaled by a power of two into fp16's range and the
    result keeps f32 accuracy; rows 1e-6 of the maximum still come out to f32 accuracy relative to their own size."""
    rng = np.random.RandomState(7)
    K, M, N = 6000, 256, 256
    a = (rng.randn(K, M) * magnitude).astype(np.float32)
    a[:, :16] *= 1e-6                                           # sixteen output rows far below the maximum
    b = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    want = a.T.astype(np.float64) @ b.astype(np.float64)
    amax = ops.absmax(dev(a))
    assert np.isclose(amax.item(), np.abs(a).max())
    got = ops.gemm(dev(a), dev(b), True, False, split_fp16=True, a_absmax=amax).cpu().numpy()
    f32 = ops.gemm(dev(a), dev(b), True, False).cpu().numpy()
    for rows in (slice(0, 16), slice(16, None)):
        scale = np.abs(want[rows]).max()
        e32, e16 = np.abs(f32[rows] - want[rows]).max() / scale, np.abs(got[rows] - want[rows]).max() / scale
        assert e16 <= 1.5 * e32 + 1e-7, (rows, e16, e32)


@pytest.mark.parametrize("M,N,K", [(1024, 256, 32 * 1000), (128, 512, 32 * 7), (256, 256, 64), (128, 256, 32), (384, 768, 32 * 1001)])
def test_gemm_split_fp16_dw_kernel_equals_the_general_kernel(ops, M, N, K):
    """The dW shape (both operands contiguous along their rows, whole 32-row slabs, 256-column tiles) runs on gemm_h3_tn8_kernel:
    bit for bit the general kernel's result (UAV_DEBUG_GEMM_TN_OFF), with and without split-K, odd and even slab counts, a
    single slab, block-scaled A, accumulate; and f32-accurate against f64."""
    rng = np.random.RandomState(M + N + K)
    a = (rng.randn(K, M) * 3e-5).astype(np.float32)
    a[:, : M // 8] *= 1e-4
    b = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    da, db = dev(a), dev(b)
    amax = ops.absmax(da)
    c0 = dev(rng.randn(M, N).astype(np.float32) * 1e-4)
    got = ops.gemm(da, db, True, False, split_fp16=True, a_absmax=amax)
    acc = ops.gemm(da, db, True, False, out=c0.clone(), accumulate=True, split_fp16=True, a_absmax=amax)
    ops.set_debug_flags("gemm_tn_off")
    try:
        ref = ops.gemm(da, db, True, False, split_fp16=True, a_absmax=amax)
        ref_acc = ops.gemm(da, db, True, False, out=c0.clone(), accumulate=True, split_fp16=True, a_absmax=amax)
    finally:
        ops.set_debug_flags()
    assert torch.equal(got, ref) and torch.equal(acc, ref_acc)
    want = a.T.astype(np.float64) @ b.astype(np.float64)
    f32 = ops.gemm(da, db, True, False).cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got.cpu().numpy() - want).max() / scale <= 1.5 * np.abs(f32 - want).max() / scale + 1e-7


def test_gemm_split_fp16_refuses_other_shapes(ops):
    a, b = torch.zeros(100, 64, device=DEV), torch.zeros(64, 128, device=DEV)
    with pytest.raises(RuntimeError, match="not supported"):
        ops.gemm(a, b, split_fp16=True)
    with pytest.raises(RuntimeError, match="not supported"):
        ops.gemm(torch.zeros(128, 64, device=DEV), torch.zeros(64, 100, device=DEV), split_fp16=True)


def flat_from_state_dict(p):
    """reference state_dict -> the flat layout of csrc/mlp.hip."""
    order = ["feature.0.weight", "feature.0.bias", "feature.1.weight", "feature.1.bias", "feature.3.weight",
             "feature.3.bias", "feature.4.weight", "feature.4.bias"]
    parts = [p[k].reshape(-1) for k in order]
    parts += [p["actor.weight"].reshape(-1), p["critic.weight"].reshape(-1), p["actor.bias"], p["critic.bias"]]
    return torch.cat(parts)


def grads_to_state_dict(flat):
    f = flat.cpu()
    o = 0
    out = {}
    for k, shape in (("feature.0.weight", (256, 6)), ("feature.0.bias", (256,)), ("feature.1.weight", (256,)),
                     ("feature.1.bias", (256,)), ("feature.3.weight", (128, 256)), ("feature.3.bias", (128,)),
                     ("feature.4.weight", (128,)), ("feature.4.bias", (128,)), ("actor.weight", (5, 128)),
                     ("critic.weight", (1, 128)), ("actor.bias", (5,)), ("critic.bias", (1,))):
        n = int(np.prod(shape))
        out[k] = f[o:o + n].reshape(shape)
        o += n
    assert o == f.numel()
    return out


@pytest.mark.parametrize("B", [1, 64, 257, 5000])
def test_mlp_fwd_bwd_vs_oracle(ops, golden, B):
    g = golden("policy_update.npz")
    p = {k: torch.from_numpy(g["init/" + k].copy()) for k in po.MLP_KEYS}
    # make LayerNorm affine non-trivial so their gradients are exercised
    torch.manual_seed(B)
    for k in ("feature.1.weight", "feature.1.bias", "feature.4.weight", "feature.4.bias"):
        p[k] = p[k] + 0.1 * torch.randn_like(p[k])
    x = torch.rand(B, 6)
    if B >= 64:
        x[:64] = torch.from_numpy(g["fwd_x"])
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    probs, value, logits = po.mlp_forward(leaf, x)
    dheads = torch.randn(B, 6) / B
    (torch.cat([logits, value], 1) * dheads).sum().backward()

    flat = flat_from_state_dict(p).to(DEV)
    heads, stash = ops.mlp_fwd(flat, x.to(DEV))
    assert torch.allclose(heads[:, :5].cpu(), logits.detach(), atol=2e-6, rtol=1e-5)
    assert torch.allclose(heads[:, 5:].cpu(), value.detach(), atol=5e-6, rtol=1e-5)
    grad = ops.mlp_bwd(flat, x.to(DEV), stash, dheads.to(DEV))
    got = grads_to_state_dict(grad)
    for k in po.MLP_KEYS:
        want = leaf[k].grad
        scale = want.abs().max().item() + 1e-12
        assert torch.allclose(got[k], want, rtol=1e-3, atol=2e-5 * scale), (k, (got[k] - want).abs().max().item(), scale)


def test_mlp_forward_golden(ops, golden):
    """Reference PPOActorCritic.forward outputs (model.py:42-53) reproduced by the HIP path."""
    g = golden("policy_update.npz")
    p = {k: torch.from_numpy(g["init/" + k].copy()) for k in po.MLP_KEYS}
    heads, _ = ops.mlp_fwd(flat_from_state_dict(p).to(DEV), dev(g["fwd_x"]))
    probs = torch.softmax(heads[:, :5], -1).cpu().numpy()
    assert np.allclose(probs, g["fwd_probs"], atol=1e-6)
    assert np.allclose(heads[:, 5:].cpu().numpy(), g["fwd_value"], atol=5e-6)


@pytest.mark.parametrize("n,H", [(64, 64), (300, 128), (5000, 128), (1000, 256)])
def test_ppo_loss_fused_heads_equals_unfused(ops, n, H):
    """uav_ppo_loss_from_y (heads by MFMA inside the loss kernel) == heads GEMM + uav_ppo_loss."""
    rng = np.random.RandomState(n + H)
    y = rng.randn(n, H).astype(np.float32)
    w = (rng.randn(6, H) * 0.2).astype(np.float32)
    b = (rng.randn(6) * 0.1).astype(np.float32)
    _, value, act, lpo, adv, ret, vo = _loss_inputs(n, n)
    heads = (y.astype(np.float64) @ w.T.astype(np.float64) + b).astype(np.float32)
    s1 = torch.empty(4, dtype=torch.float64, device=DEV)
    d1 = torch.empty(n, 6, device=DEV)
    db1 = torch.empty(6, device=DEV)
    ops.ppo_loss_heads(dev(heads), dev(act), dev(lpo), dev(adv), dev(ret), dev(vo), 1.0 / n, 0.2, 0.01, s1, d1, db1)
    s2 = torch.empty(4, dtype=torch.float64, device=DEV)
    d2 = torch.empty(n, 6, device=DEV)
    db2 = torch.empty(6, device=DEV)
    ops.ppo_loss_from_y(dev(y), dev(w), dev(b), dev(act), dev(lpo), dev(adv), dev(ret), dev(vo), 1.0 / n, 0.2, 0.01, s2, d2, db2)
    assert np.allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert np.allclose(d2.cpu().numpy(), d1.cpu().numpy(), rtol=2e-4, atol=1e-6 / n)
    assert np.allclose(db2.cpu().numpy(), db1.cpu().numpy(), rtol=1e-3, atol=1e-6)
    assert np.allclose(db2.cpu().numpy(), d2.cpu().numpy().sum(0), rtol=1e-3, atol=1e-6)


def test_pack_success_bits_matches_bruteforce(ops):
    """uav_pack_success_bits: order-preserving compaction of the ended episodes' success bits, count header, capacity clamp."""
    rng = np.random.RandomState(0)
    for n, p, cap in ((1, 1.0, 8), (1000, 0.02, 64), (4096 * 128, 0.0015, 16384), (70001, 0.3, 5000), (513, 0.0, 16)):
        ended = rng.rand(n) < p
        flags = (ended * np.where(rng.rand(n) < 0.5, 3, 1)).astype(np.uint8)
        msg = ops.pack_success_bits(torch.from_numpy(flags).to(DEV), cap).cpu().numpy()
        cnt = int(msg[0]) | int(msg[1]) << 8 | int(msg[2]) << 16 | int(msg[3]) << 24
        want = (flags[ended] >> 1) & 1
        assert cnt == ended.sum() and msg.shape == (4 + cap + 1,)
        k = min(cnt, cap)
        assert np.array_equal(msg[4:4 + k], want[:k]) and not msg[4 + k:].any()


def test_device_curriculum_matches_the_reference_traces(golden):
    """uav_curriculum_update (PPOTrainer.update, model.py:131-164, on the device) against the reference's own traces
    (tests/golden/curriculum.npz): fed one episode per call, radius / bonus after every episode to 1e-12 (device pow vs libm),
    the np.float64 switch of the bonus at the first full window; fed in random chunks through uav_pack_success_bits' message
    format over three 'ranks', the same final state; counters and window length as the host class."""
    from uavppo import ops
    from uavppo.curriculum import Curriculum
    g = golden("curriculum.npz")
    cap = 64
    rng = np.random.RandomState(3)

    def message(bits):
        m = np.zeros(4 + cap + 1, np.uint8)
        m[:4] = np.frombuffer(np.int32(len(bits)).tobytes(), np.uint8)
        m[4:4 + len(bits)] = np.asarray(bits, np.uint8)
        return m
    for name in sorted({k.split("/")[0] for k in g.files if "/" in k}):
        seq, trace = g[f"{name}/seq"], g[f"{name}/trace"]
        st = ops.curriculum_state(DEV)
        host = Curriculum()
        for i, s in enumerate(seq[:400]):
            ops.curriculum_update(st, torch.from_numpy(message([int(s)])[None]).to(DEV), cap)
            host.update(bool(s))
            if i % 7 == 0 or i + 1 == min(len(seq), 400):
                d = ops.curriculum_read(st.cpu().numpy())
                assert np.allclose([d["radius"], d["bonus"]], trace[i][:2], rtol=1e-12, atol=0), (name, i)
                assert d["bonus_is_f64"] == isinstance(host.explore_bonus, np.float64)
                assert d["hist_len"] == len(host.success_history) and d["episodes"] == i + 1
        # the whole sequence in chunks over three message rows per call
        st2 = ops.curriculum_state(DEV)
        i = 0
        while i < len(seq):
            rows = []
            for _ in range(3):
                k = int(rng.randint(0, cap + 1))
                rows.append(message([int(v) for v in seq[i:i + k]]))
                i += len(seq[i:i + k])
            ops.curriculum_update(st2, torch.from_numpy(np.stack(rows)).to(DEV), cap)
        d2 = ops.curriculum_read(st2.cpu().numpy())
        assert np.allclose([d2["radius"], d2["bonus"]], trace[len(seq) - 1][:2], rtol=1e-12, atol=0), name
        assert d2["episodes"] == len(seq) and d2["successes"] == int(np.sum(seq)) and not d2["overflow"]
    # messages as long as a 4096-env rollout's (windows of 120 inside one message, several 4096-byte staging chunks, windows that
    # straddle chunk and rank boundaries): the device state after each call equals the host class fed episode by episode
    big = 16384
    st4, host4 = ops.curriculum_state(DEV), Curriculum()
    for call in range(4):
        rows = np.zeros((3, 4 + big + 1), np.uint8)
        for r, n in enumerate((int(rng.randint(4000, 9000)), int(rng.randint(0, 300)), int(rng.randint(9000, big + 1)))):
            bits = (rng.rand(n) < (0.15 + 0.2 * call)).astype(np.uint8)
            rows[r, :4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
            rows[r, 4:4 + n] = bits
            for b in bits:
                host4.update(bool(b))
        ops.curriculum_update(st4, torch.from_numpy(rows).to(DEV), big)
        d4 = ops.curriculum_read(st4.cpu().numpy())
        assert np.allclose([d4["radius"], d4["bonus"]], [host4.current_radius, float(host4.explore_bonus)], rtol=1e-12, atol=0), call
        assert d4["hist_len"] == len(host4.success_history) and not d4["overflow"]
    # more episodes in a message than it holds: flagged
    st3 = ops.curriculum_state(DEV)
    m = message([1] * cap)
    m[:4] = np.frombuffer(np.int32(cap + 5).tobytes(), np.uint8)
    ops.curriculum_update(st3, torch.from_numpy(m[None]).to(DEV), cap)
    assert ops.curriculum_read(st3.cpu().numpy())["overflow"]
