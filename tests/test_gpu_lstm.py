"""HIP LSTM sequence kernels vs torch.nn.LSTM (CPU, the module the reference uses:
PPOV2.0/model.py:206-212, PPOV2.1/model.py:263) and the oracle's episode-reset variant.  -m gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from uavppo import ops as o
    return o


def _stash_equal(a, b):
    """The BPTT stash rows [.., 6H] = gates i, f, g, o | c_prev | h_prev.  The h_prev slot is part of the contract only where the
    weight-gradient pass reads it from there (the f32-rows forms); with I <= 6 at h = 64 / 128 and on the h = 256 fp16-split path
    (h_prev = y one step back under the restart mask, csrc/wgrad_pc.hip) it is not written at all."""
    H5 = a.shape[-1] // 6 * 5
    return torch.equal(a[..., :H5], b[..., :H5])


def _close(a, b, rtol=2e-4, atol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= atol * max(scale, 1.0) + rtol * scale, (err, scale)


@pytest.mark.parametrize("T,N,I,H", [(8, 4, 6, 64), (16, 8, 6, 128), (70, 37, 6, 128), (5, 16, 8, 64), (33, 300, 4, 64),
                                     (128, 64, 6, 128), (64, 128, 6, 64),      # shapes that take the LDS-DMA wgrad/bwd variants
                                     (12, 20, 64, 64), (9, 17, 128, 128), (1, 1, 6, 128)])
@pytest.mark.parametrize("use_keep", [False, True])
def test_lstm_layer_fwd_bwd(ops, T, N, I, H, use_keep):
    torch.manual_seed(T * 100 + N)
    ref = torch.nn.LSTM(I, H, 1)
    w_ih, w_hh, b_ih, b_hh = [p.detach().clone().requires_grad_(True) for p in ref.parameters()]
    x = torch.randn(T, N, I, requires_grad=True)
    h0 = torch.randn(N, H, requires_grad=True)
    c0 = torch.randn(N, H, requires_grad=True)
    keep = (torch.rand(T, N) > 0.2).float() if use_keep else None
    if use_keep:
        keep[0, 0] = 0.0
    y, hn, cn = po.lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, keep)
    if not use_keep:                      # restatement == torch.nn.LSTM itself
        y2, (hn2, cn2) = ref(x.detach(), (h0.detach()[None], c0.detach()[None]))
        assert torch.allclose(y, y2, atol=1e-6) and torch.allclose(cn, cn2[0], atol=1e-6)
    dy = torch.randn(T, N, H)
    dhn, dcn = torch.randn(N, H), torch.randn(N, H)
    ((y * dy).sum() + (hn * dhn).sum() + (cn * dcn).sum()).backward()

    d = lambda t: t.detach().to(DEV).contiguous()
    xg = d(x.transpose(0, 1))                      # (env, T, feat)
    kg = d(keep.transpose(0, 1)) if use_keep else None
    yg, hng, cng, stash = ops.lstm_fwd(xg, kg, d(h0), d(c0), d(w_ih), d(w_hh), d(b_ih), d(b_hh))
    _close(yg.transpose(0, 1), y, 1e-5, 2e-6)
    _close(hng, hn, 1e-5, 2e-6)
    _close(cng, cn, 1e-5, 2e-6)
    g = ops.lstm_bwd(xg, kg, stash, d(w_ih), d(w_hh), yg, d(h0), dy=d(dy.transpose(0, 1)), dhn=d(dhn), dcn=d(dcn),
                     need_dx=True)
    _close(g["dx"].transpose(0, 1), x.grad)
    _close(g["dw_ih"], w_ih.grad)
    _close(g["dw_hh"], w_hh.grad)
    _close(g["db"], b_ih.grad)
    _close(g["dh0"], h0.grad)
    _close(g["dc0"], c0.grad)


def test_lstm_bwd_fused_heads_equals_explicit_dy(ops):
    """dheads + w_head path (dy formed in registers) == passing dy = dheads @ w_head."""
    torch.manual_seed(3)
    N, T, I, H = 33, 40, 6, 128
    dev = DEV
    x = torch.randn(N, T, I, device=dev)
    keep = (torch.rand(N, T, device=dev) > 0.1).float()
    w_ih, w_hh = torch.randn(4 * H, I, device=dev) * 0.1, torch.randn(4 * H, H, device=dev) * 0.1
    b = torch.randn(4 * H, device=dev) * 0.1
    h0, c0 = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
    y, hn, cn, stash = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b, b)
    dheads = torch.randn(N, T, 6, device=dev)
    w_head = torch.randn(6, H, device=dev)
    dy = (dheads.reshape(-1, 6).cpu().double() @ w_head.cpu().double()).float().reshape(N, T, H).to(dev)
    g1 = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dy=dy)
    g2 = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dheads=dheads, w_head=w_head)
    for k in ("dw_ih", "dw_hh", "db", "dh0", "dc0"):
        _close(g2[k], g1[k], 1e-4, 1e-5)
    # fused head-weight gradient dW_head = dheads^T y
    want = dheads.reshape(-1, 6).cpu().double().T @ y.reshape(-1, H).cpu().double()
    _close(g2["dw_head"], want.float(), 1e-4, 1e-5)


def test_lstm_two_layer_stack_matches_torch(ops):
    """Stacked layers (BASELINE config C5 'h=256 stacked x2' shape family, here H=128): layer 2 takes
    the time-batched input-projection path (I = H > 8)."""
    torch.manual_seed(5)
    T, N, I, H = 10, 19, 8, 128
    ref = torch.nn.LSTM(I, H, 2)
    x = torch.randn(T, N, I)
    h0, c0 = torch.randn(2, N, H), torch.randn(2, N, H)
    y_ref, (hn_ref, cn_ref) = ref(x, (h0, c0))
    d = lambda t: t.detach().to(DEV).contiguous()
    p = {k: d(v) for k, v in ref.named_parameters()}
    y1, hn1, cn1, _ = ops.lstm_fwd(d(x.transpose(0, 1)), None, d(h0[0]), d(c0[0]), p["weight_ih_l0"], p["weight_hh_l0"],
                                   p["bias_ih_l0"], p["bias_hh_l0"])
    y2, hn2, cn2, _ = ops.lstm_fwd(y1, None, d(h0[1]), d(c0[1]), p["weight_ih_l1"], p["weight_hh_l1"],
                                   p["bias_ih_l1"], p["bias_hh_l1"])
    _close(y2.transpose(0, 1), y_ref, 1e-5, 3e-6)
    _close(hn2, hn_ref[1], 1e-5, 3e-6)
    _close(cn1, cn_ref[0], 1e-5, 3e-6)


@pytest.mark.parametrize("T,N,I,H", [(6, 5, 6, 256), (9, 33, 256, 256), (4, 3, 6, 32)])
def test_lstm_generic_hidden_sizes(ops, T, N, I, H):
    """Hidden sizes without a persistent kernel (BASELINE C5: h=256) run the per-step GEMM + pointwise
    path behind the same entry points; same parity bar against torch.nn.LSTM semantics."""
    torch.manual_seed(H + T)
    ref = torch.nn.LSTM(I, H, 1)
    w_ih, w_hh, b_ih, b_hh = [p.detach().clone().requires_grad_(True) for p in ref.parameters()]
    x = torch.randn(T, N, I, requires_grad=True)
    h0 = torch.randn(N, H, requires_grad=True)
    c0 = torch.randn(N, H, requires_grad=True)
    keep = (torch.rand(T, N) > 0.25).float()
    y, hn, cn = po.lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, keep)
    dy, dhn, dcn = torch.randn(T, N, H), torch.randn(N, H), torch.randn(N, H)
    ((y * dy).sum() + (hn * dhn).sum() + (cn * dcn).sum()).backward()
    d = lambda t: t.detach().to(DEV).contiguous()
    xg, kg = d(x.transpose(0, 1)), d(keep.transpose(0, 1))
    yg, hng, cng, stash = ops.lstm_fwd(xg, kg, d(h0), d(c0), d(w_ih), d(w_hh), d(b_ih), d(b_hh))
    _close(yg.transpose(0, 1), y, 1e-5, 3e-6)
    _close(cng, cn, 1e-5, 3e-6)
    g = ops.lstm_bwd(xg, kg, stash, d(w_ih), d(w_hh), yg, d(h0), dy=d(dy.transpose(0, 1)), dhn=d(dhn), dcn=d(dcn), need_dx=True)
    for k, want in (("dx", x.grad.transpose(0, 1)), ("dw_ih", w_ih.grad), ("dw_hh", w_hh.grad), ("db", b_ih.grad),
                    ("dh0", h0.grad), ("dc0", c0.grad)):
        _close(g[k], want)


@pytest.mark.parametrize("split", ["fp16x3", "bf16x6"])
@pytest.mark.parametrize("N,T,H", [(64, 32, 128), (128, 16, 64), (48, 64, 128)])
def test_split_kernels_have_f32_accuracy(ops, N, T, H, split):
    """The default LSTM kernels multiply on the 16-bit matrix pipe: forward / backward with a two-piece fp16 operand split
    and three products (common.h split2h), the weight gradients -- and everything under UAV_ARITH_BF16X6 -- with a
    three-piece bf16 split and six products.  Claim: the result is an f32 computation -- against an f64 LSTM their error
    is that of the exact-f32-MFMA kernels (UAV_ARITH_F32_MFMA selects those), for forward, backward and weight
    gradients (shapes with full 32-row slabs)."""
    torch.manual_seed(N + T)
    I, A = 6, 6
    ref = torch.nn.LSTM(I, H, 1).double()
    w_ih, w_hh, b_ih, b_hh = [p.detach().clone().requires_grad_(True) for p in ref.parameters()]
    x = torch.randn(T, N, I, dtype=torch.float64)
    h0 = (torch.randn(N, H, dtype=torch.float64) * 0.3).requires_grad_(True)
    c0 = (torch.randn(N, H, dtype=torch.float64) * 0.3).requires_grad_(True)
    keep = (torch.rand(T, N) > 0.1).double()
    keep[0, 0] = 0.0
    w_head = (torch.randn(A, H, dtype=torch.float64) * 0.2).requires_grad_(True)
    dheads = torch.randn(T, N, A, dtype=torch.float64) / (N * T)
    y, hn, cn = po.lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, keep)
    ((y @ w_head.T) * dheads).sum().backward()
    want = {"y": y.transpose(0, 1), "dw_ih": w_ih.grad, "dw_hh": w_hh.grad, "db": b_ih.grad, "dh0": h0.grad,
            "dc0": c0.grad, "dw_head": w_head.grad}

    d = lambda t: t.detach().float().to(DEV).contiguous()
    xg, kg = d(x.transpose(0, 1)), d(keep.transpose(0, 1))
    args = (d(h0), d(c0), d(w_ih), d(w_hh), d(b_ih), d(b_hh))

    def run():
        yg, hng, cng, stash = ops.lstm_fwd(xg, kg, *args)
        g = ops.lstm_bwd(xg, kg, stash, args[2], args[3], yg, args[0], dheads=d(dheads.transpose(0, 1)), w_head=d(w_head))
        g["y"] = yg
        return {k: g[k].detach().cpu().double() for k in want}

    def errs(got):
        return {k: float((got[k] - want[k].detach()).abs().max() / (want[k].detach().abs().max() + 1e-30)) for k in want}

    with ops.lstm_arith(split):
        e_x6 = errs(run())
    with ops.lstm_arith("f32_mfma"):
        e_f32 = errs(run())
    assert ops.get_lstm_arith() == "fp16x3"
    for k in want:
        assert e_x6[k] < 5e-6, (k, e_x6)                                 # f32-level agreement with the f64 reference
        assert e_x6[k] <= 2.0 * e_f32[k] + 2e-7, (k, e_x6[k], e_f32[k])   # ... and no worse than the exact-f32 MFMA chain


@pytest.mark.parametrize("H", [128, 64])
def test_split_fp16_backward_keeps_f32_accuracy_over_40_decades(ops, H):
    """fp16 has 5 exponent bits, gradients do not care: the backward scales each env's gate gradients by a power of two
    per step before the split.  Envs whose loss gradients differ by up to 1e25 (and vary by 1e6 along their own
    sequence) must each come out with f32 relative accuracy -- measured per env against an f64 LSTM."""
    torch.manual_seed(H)
    N, T, I, A = 40, 24, 6, 6
    ref = torch.nn.LSTM(I, H, 1).double()
    w_ih, w_hh, b_ih, b_hh = [p.detach().clone().requires_grad_(True) for p in ref.parameters()]
    x = torch.randn(T, N, I, dtype=torch.float64)
    h0 = (torch.randn(N, H, dtype=torch.float64) * 0.3).requires_grad_(True)
    c0 = (torch.randn(N, H, dtype=torch.float64) * 0.3).requires_grad_(True)
    w_head = torch.randn(A, H, dtype=torch.float64) * 0.2
    env_scale = 10.0 ** torch.linspace(-25, 0, N, dtype=torch.float64)
    time_scale = 10.0 ** (-6.0 * torch.rand(T, 1, dtype=torch.float64))
    dheads = torch.randn(T, N, A, dtype=torch.float64) * env_scale[None, :, None] * time_scale[:, :, None]
    y, hn, cn = po.lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, None)
    ((y @ w_head.T) * dheads).sum().backward()
    d = lambda t: t.detach().float().to(DEV).contiguous()
    xg = d(x.transpose(0, 1))
    yg, _, _, stash = ops.lstm_fwd(xg, None, d(h0), d(c0), d(w_ih), d(w_hh), d(b_ih), d(b_hh))
    g = ops.lstm_bwd(xg, None, stash, d(w_ih), d(w_hh), yg, d(h0), dheads=d(dheads.transpose(0, 1)), w_head=d(w_head))
    for k, want in (("dh0", h0.grad), ("dc0", c0.grad)):
        got = g[k].cpu().double()
        rel = (got - want).abs().amax(1) / want.abs().amax(1)                 # per env, against its own magnitude
        assert float(rel.max()) < 5e-6, (k, rel)
        assert float(want.abs().amax(1).min()) < 1e-20 < 1e-4 < float(want.abs().amax(1).max())   # the span is real


@pytest.mark.parametrize("N,T,I,H", [(37, 21, 6, 128), (16, 9, 6, 64), (5, 1, 6, 128), (7, 6, 6, 256), (9, 12, 64, 128)])
@pytest.mark.parametrize("f32_mfma", [False, True])
def test_lstm_fwd_emits_heads(ops, N, T, I, H, f32_mfma):
    """uav_lstm_fwd's heads output == (y W_head^T + b_head) of model.py:44,52, whether the sequence kernel forms it
    itself (split-bf16 kernels, from the unmasked h planes: episode resets inside the sequence must not leak into it)
    or the entry point falls back to one GEMM over y (exact-f32 / wide-input / h=256 paths)."""
    ops.set_lstm_arith("f32_mfma" if f32_mfma else "fp16x3")
    torch.manual_seed(N * 7 + T)
    dev = DEV
    x = torch.randn(N, T, I, device=dev)
    keep = (torch.rand(N, T, device=dev) > 0.3).float()
    k = 1.0 / H ** 0.5
    w_ih, w_hh = (torch.rand(4 * H, I, device=dev) * 2 - 1) * k, (torch.rand(4 * H, H, device=dev) * 2 - 1) * k
    b_ih, b_hh = (torch.rand(4 * H, device=dev) * 2 - 1) * k, (torch.rand(4 * H, device=dev) * 2 - 1) * k
    h0, c0 = torch.randn(N, H, device=dev) * 0.3, torch.randn(N, H, device=dev) * 0.3
    w_head, b_head = torch.randn(6, H, device=dev) * 0.3, torch.randn(6, device=dev)
    heads = torch.full((N, T, 6), float("nan"), device=dev)
    y, hn, cn, stash = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, w_head=w_head, b_head=b_head, heads=heads)
    want = y.cpu().double() @ w_head.cpu().double().T + b_head.cpu().double()
    _close(heads, want.float(), 1e-5, 2e-6)
    # and y itself is unchanged by asking for heads
    y2, _, _, _ = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh)
    assert torch.equal(y, y2)
    ops.set_lstm_arith("fp16x3")


def test_lstm_bwd_fused_path_fuzz_against_dy_path(ops):
    """Many small ragged shapes: the PPO backward (dheads + w_head: split-bf16 K-split kernel, LDS-DMA stash ring)
    against the plain-dy exact-f32 kernel fed dy = dheads @ w_head -- recurrent gradients, weight gradients, initial-state
    gradients.  Catches indexing / ring / partial-sum mistakes at sizes where an error cannot hide in tolerance."""
    rng = np.random.RandomState(7)
    dev = DEV
    for case in range(24):
        H = (64, 128)[case % 2]
        N, T = int(rng.randint(1, 70)), int(rng.randint(1, 45))
        torch.manual_seed(case)
        x = torch.randn(N, T, 6, device=dev)
        keep = (torch.rand(N, T, device=dev) > 0.15).float() if case % 3 else None
        w_ih, w_hh = torch.randn(4 * H, 6, device=dev) * 0.1, torch.randn(4 * H, H, device=dev) * 0.1
        b = torch.randn(4 * H, device=dev) * 0.1
        h0, c0 = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
        y, hn, cn, stash = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b, b)
        dheads = torch.randn(N, T, 6, device=dev)
        w_head = torch.randn(6, H, device=dev)
        dy = (dheads.reshape(-1, 6).cpu().double() @ w_head.cpu().double()).float().reshape(N, T, H).to(dev)
        g1 = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dy=dy)
        g2 = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dheads=dheads, w_head=w_head)
        for k in ("dgates", "dw_ih", "dw_hh", "db", "dh0", "dc0"):
            a, bb = g2[k].double().cpu(), g1[k].double().cpu()
            scale = bb.abs().max().item() + 1e-12
            assert (a - bb).abs().max().item() <= 2e-5 * max(scale, 1.0) + 1e-4 * scale, (case, N, T, H, k)


@pytest.mark.parametrize("H", [128, 64])
def test_split_fp16_weight_gradients_over_a_wide_dynamic_range(ops, H):
    """The weight-gradient kernel's fp16 split keeps ONE accumulator per tile (unscaled residuals) under a running
    power-of-two scale PER GATE ROW that is lowered, and that row of the accumulators rescaled, when larger gradients
    arrive.  Checked against f64 sums: (a) gate rows up to 2^31 below the largest row of their wave all come out at f32
    relative accuracy, (c) gradients that grow by 2^40 along the samples (many rescales) and (d) that start with all-zero
    slabs are handled.  (A single scale per wave, tried first, left rows 2^24 below the largest at 6e-5 and rows 2^31
    below at 1e-2 of their own magnitude.)"""
    torch.manual_seed(H + 1)
    N, T, I = 32, 64, 6
    g = torch.Generator().manual_seed(H)
    x = torch.rand(N, T, I, generator=g, dtype=torch.float64)
    y = torch.rand(N, T, H, generator=g, dtype=torch.float64) * 2 - 1
    h0 = torch.rand(N, H, generator=g, dtype=torch.float64) * 2 - 1
    keep = (torch.rand(N, T, generator=g) > 0.1).double()
    dheads = torch.randn(N, T, 6, generator=g, dtype=torch.float64) * 1e-3
    hprev = torch.cat([h0[:, None], y[:, :-1]], 1) * keep[..., None]
    stash = torch.zeros(N, T, 6 * H)                                         # the I <= 6 fast path does not read it
    d = lambda t: t.float().to(DEV).contiguous()

    def check(dg, tag, row_tol):
        want_hh = torch.einsum("ntm,ntu->mu", dg, hprev)
        want_ih = torch.einsum("ntm,nti->mi", dg, x)
        want_b = dg.sum((0, 1))
        got = ops.lstm_wgrad(d(x), d(keep), d(h0), d(y), stash.to(DEV), d(dg), torch.zeros(4 * H, I, device=DEV), dheads=d(dheads))
        dgf, hpf, xf = dg.float().double(), hprev.float().double(), x.float().double()      # what the kernel is given
        want_hh, want_ih, want_b = (torch.einsum("ntm,ntu->mu", dgf, hpf), torch.einsum("ntm,nti->mi", dgf, xf), dgf.sum((0, 1)))
        big = float(want_hh.abs().max())
        for name, w_, g_ in (("dw_hh", want_hh, got["dw_hh"]), ("dw_ih", want_ih, got["dw_ih"])):
            err = (g_.cpu().double() - w_).abs()
            assert float(err.max()) <= 5e-7 * float(w_.abs().max()), (tag, name, float(err.max()), big)
            rel = err.amax(1) / w_.abs().amax(1).clamp_min(1e-300)            # per gate row, against its own magnitude
            for lo, hi, tol in row_tol:
                assert float(rel[lo:hi].max()) < tol, (tag, name, lo, hi, float(rel[lo:hi].max()))
        # the bias gradient is a single signed sum per row: measured against the sum of magnitudes (no cancellation)
        errb = (got["db"].cpu().double() - want_b).abs() / dgf.abs().sum((0, 1)).clamp_min(1e-300)
        assert float(errb.max()) < 3e-7, (tag, "db", float(errb.max()))
        want_head = torch.einsum("nta,ntu->au", dheads.float().double(), y.float().double())
        assert float((got["dw_head"].cpu().double() - want_head).abs().max()) < 3e-6 * float(want_head.abs().max())

    base = torch.randn(N, T, 4 * H, generator=g, dtype=torch.float64) * 1e-5
    # (a)+(b): within every wave's 64 gate rows, row r is scaled by 2^-(r % 32) ... 2^-31
    scale = 2.0 ** -(torch.arange(4 * H) % 32).double()
    check(base * scale, "rows", [(0, 4 * H, 5e-6)])          # every row has its own scale: 31 binades below the largest changes nothing
    # (c): magnitudes growing by 2^40 over the row index (n, t) -> the running scale is lowered again and again
    grow = 2.0 ** (40.0 * torch.arange(N * T).double() / (N * T) - 40.0).reshape(N, T, 1)
    check(base * grow, "growing", [(0, 4 * H, 5e-6)])
    # (d): the first quarter of the rows is exactly zero
    z = base.clone()
    z[: N // 4] = 0.0
    check(z, "zeros first", [(0, 4 * H, 5e-6)])
    # (e): inputs far from O(1) (the x | 1 columns carry their own, smaller block scale: |x| < 4096)
    x.mul_(3000.0)
    check(base, "large x", [(0, 4 * H, 5e-6)])


def test_gate_activation_error_bounds(ops):
    """csrc/common.h: fast_sigmoid / fast_tanh are v_exp_f32 + v_rcp_f32 (no IEEE division sequence).  tanh is 1 - 2/(exp(2x)+1)
    for |x| >= 1/4 (absolute error <= 2.5e-7) and, since round 5, the odd Taylor polynomial to x^9 below (the exp form cancels
    there: its relative error was 2.5e-7 / |x|, 1e-3 at |x| = 1e-4): RELATIVE error <= 1.5e-6 over the whole range, down to
    |x| = 1e-6, stated in DESIGN.md 3 and pinned here.
    The activations are read straight from the BPTT stash (gates after activation) of a one-step layer with W = 0."""
    H, N = 128, 64
    xs = torch.cat([torch.linspace(-12, 12, 4096), torch.logspace(-6, 0, 2048), -torch.logspace(-6, 0, 2048)]).double()
    xs = xs[: (xs.numel() // H) * H].reshape(-1, H)                     # one row of pre-activations per "env"
    rows = xs.shape[0]
    w_ih = torch.zeros(4 * H, 6, device=DEV)
    w_ih[:, 0] = 1.0                                                      # gate pre-activation = x[0] (+ bias 0)
    w_hh = torch.zeros(4 * H, H, device=DEV)
    b = torch.zeros(4 * H, device=DEV)
    worst_abs_t = worst_abs_s = worst_rel_t = worst_relx_t = 0.0
    for r in range(rows):
        # every unit of env n sees the same scalar; put value xs[r, u] on env u (N = H envs)
        x = torch.zeros(H, 1, 6, device=DEV)
        x[:, 0, 0] = xs[r].float().to(DEV)
        z = torch.zeros(H, H, device=DEV)
        _, _, _, stash = ops.lstm_fwd(x, None, z, z, w_ih, w_hh, b, b)
        st = stash[:, 0].cpu().double()
        pre = xs[r].float().double()[:, None]                             # the f32 value the kernel saw
        sig, tnh = st[:, 0:H], st[:, 2 * H:3 * H]                        # i gate (sigmoid), g gate (tanh)
        worst_abs_s = max(worst_abs_s, (sig - torch.sigmoid(pre)).abs().max().item())
        et = (tnh - torch.tanh(pre)).abs()
        worst_abs_t = max(worst_abs_t, et.max().item())
        worst_rel_t = max(worst_rel_t, (et / torch.tanh(pre).abs().clamp_min(1e-300)).max().item())     # (x = 0: 0 / tiny = 0)
        worst_relx_t = max(worst_relx_t, (et * pre.abs().clamp_max(1.0) / torch.tanh(pre).abs().clamp_min(1e-300)).max().item())
    print("sigmoid abs", worst_abs_s, "tanh abs", worst_abs_t, "tanh rel", worst_rel_t, "tanh rel*min(|x|,1)", worst_relx_t)
    assert worst_abs_s < 1.5e-7 and worst_abs_t < 2.5e-7
    assert worst_rel_t < 1.5e-6            # relative error of tanh over [1e-6, 12]: largest just above the 1/4 switch (2.5e-7 / 0.245); the exp form alone: 1e-3 at 1e-4


@pytest.mark.parametrize("N,T,I", [(64, 6, 8), (100, 5, 256), (37, 3, 40)])
def test_lstm_stepper_equals_sequence_forward(ops, N, T, I):
    """uav_lstm_stepper_* (one time step per call: the step-wise rollout of an h = 256 policy) runs the kernels of
    uav_lstm_fwd: given each step's restart mask, y, the BPTT stash and the final state
    are BIT-identical to the sequence call over the same inputs -- which is why PPO epoch 0 may adopt a rollout's stash."""
    H = 256
    g = torch.Generator().manual_seed(N + T + I)
    x = torch.randn(N, T, I, generator=g).to(DEV)
    keep = (torch.rand(N, T, generator=g) > 0.25).float()
    keep[:, 0] = 1.0
    keep = keep.to(DEV)
    h0, c0 = (torch.randn(N, H, generator=g) * 0.5).to(DEV), (torch.randn(N, H, generator=g) * 0.5).to(DEV)
    w_ih, w_hh = (torch.randn(4 * H, I, generator=g) * 0.1).to(DEV), (torch.randn(4 * H, H, generator=g) * 0.1).to(DEV)
    b_ih, b_hh = (torch.randn(4 * H, generator=g) * 0.1).to(DEV), (torch.randn(4 * H, generator=g) * 0.1).to(DEV)
    y_ref, hn_ref, cn_ref, stash_ref = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh)
    sp = ops.LstmStepper(N, I, H, DEV)
    sp.begin(w_ih, w_hh, b_ih, b_hh, h0, c0)
    y, stash = torch.zeros(N, T, H, device=DEV), torch.zeros(N, T, 6 * H, device=DEV)
    for t in range(T):
        sp.step(x, t, y, stash, keep=keep[:, t].contiguous())
    assert torch.equal(y, y_ref) and _stash_equal(stash, stash_ref)
    assert torch.equal(sp.hn, hn_ref) and torch.equal(sp.cn, cn_ref)
    # a second rollout from the handed-over state: begin() again, some envs restarting at step 0
    sp.begin(w_ih, w_hh, b_ih, b_hh, hn_ref, cn_ref)
    k0 = (torch.rand(N, generator=g) > 0.5).float().to(DEV)
    keep2 = keep.clone()
    keep2[:, 0] = k0
    y2_ref, _, _, st2_ref = ops.lstm_fwd(x, keep2, hn_ref, cn_ref, w_ih, w_hh, b_ih, b_hh)
    sp.step(x, 0, y, stash, keep=k0)
    assert torch.equal(y[:, 0], y2_ref[:, 0]) and _stash_equal(stash[:, 0], st2_ref[:, 0])


def test_stepper_step_pair_equals_the_two_steps(ops):
    """uav_lstm_stepper_step_pair: layer 1's step t + 1 and layer 2's step t (reading layer 1's piece planes of step t) as ONE launch --
    the update's forward pass over a stored sequence (uavppo/policy.py) -- gives the bits of the layer-by-layer stepper loop: y, the
    BPTT stash, final states, for a ragged env count and with restart masks."""
    H, N, T, I = 256, 100, 7, 8
    g = torch.Generator().manual_seed(5)
    mk = lambda *shape, s=0.1: (torch.randn(*shape, generator=g) * s).to(DEV)
    x = mk(N, T, I, s=1.0)
    keep = (torch.rand(T, N, generator=g) > 0.2).float().to(DEV)
    W = [(mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)), (mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H))]
    h0, c0 = [mk(N, H, s=0.3), mk(N, H, s=0.3)], [mk(N, H, s=0.3), mk(N, H, s=0.3)]

    def run(pairs):
        sp = [ops.LstmStepper(N, I, H, DEV), ops.LstmStepper(N, H, H, DEV)]
        for l in range(2):
            sp[l].begin(*W[l], h0[l], c0[l])
        y = [torch.zeros(N, T, H, device=DEV) for _ in range(2)]
        st = [torch.zeros(N, T, 6 * H, device=DEV) for _ in range(2)]
        if pairs:
            sp[0].step(x, 0, y[0], st[0], keep=keep[0])
            for t in range(T - 1):
                ops.lstm_stepper_step_pair((sp[0], x, t + 1, y[0], st[0], None, keep[t + 1]), (sp[1], y[0], t, y[1], st[1], sp[0], keep[t]))
            sp[1].step(y[0], T - 1, y[1], st[1], below=sp[0], keep=keep[T - 1])
        else:
            for t in range(T):
                sp[0].step(x, t, y[0], st[0], keep=keep[t])
                sp[1].step(y[0], t, y[1], st[1], below=sp[0], keep=keep[t])
        torch.cuda.synchronize()
        return y, st, [s.hn.clone() for s in sp], [s.cn.clone() for s in sp]

    a, b = run(True), run(False)
    for l in range(2):
        assert torch.equal(a[0][l], b[0][l]) and _stash_equal(a[1][l], b[1][l]), l
        assert torch.equal(a[2][l], b[2][l]) and torch.equal(a[3][l], b[3][l]), l


@pytest.mark.parametrize("mode", ["bf16x6", "f32_mfma"])
def test_h256_wide_range_modes_leave_the_fp16_step_kernels(ops, mode):
    """The h = 256 step kernels exist in the fp16-split form only (|w| < 65504, |x| < 4096).  A caller that selects a
    wide-range mode does so BECAUSE its operands left that range: both modes must take the generic exact-f32 step path
    (results finite and equal to the f32 reference with a weight of 1e5 and inputs of 1e4), uav_lstm_bwd_caps must report
    0 and the stepper must refuse -- not silently overflow to inf in the fp16 pieces."""
    T, N, I, H = 5, 9, 8, 256
    torch.manual_seed(3)
    ref = torch.nn.LSTM(I, H, 1)
    w_ih, w_hh, b_ih, b_hh = [p.detach().clone() for p in ref.parameters()]
    w_hh[5, 7] = 1.0e5                           # far outside fp16's range (the unit saturates; everything stays finite)
    w_ih[300, 2] = -7.0e4
    for w in (w_ih, w_hh, b_ih, b_hh):
        w.requires_grad_(True)
    x = torch.randn(T, N, I)
    x[:, :, 0] *= 1.0e4
    x.requires_grad_(True)
    h0 = torch.randn(N, H, requires_grad=True)
    c0 = torch.randn(N, H, requires_grad=True)
    y, hn, cn = po.lstm_layer_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh, None)
    dy = torch.randn(T, N, H)
    (y * dy).sum().backward()
    d = lambda t: t.detach().to(DEV).contiguous()
    xg = d(x.transpose(0, 1))
    with ops.lstm_arith(mode):
        assert ops.lstm_bwd_caps(DEV, I, H) == 0 and ops.lstm_bwd_caps(DEV, H, H) == 0
        yg, hng, cng, stash = ops.lstm_fwd(xg, None, d(h0), d(c0), d(w_ih), d(w_hh), d(b_ih), d(b_hh))
        g = ops.lstm_bwd(xg, None, stash, d(w_ih), d(w_hh), yg, d(h0), dy=d(dy.transpose(0, 1)), need_dx=True)
        sp = ops.LstmStepper(N, I, H, DEV)
        with pytest.raises(RuntimeError, match="fp16-split"):
            sp.begin(d(w_ih), d(w_hh), d(b_ih), d(b_hh), d(h0), d(c0))
    assert torch.isfinite(yg).all() and all(torch.isfinite(v).all() for v in g.values() if v is not None)
    _close(yg.transpose(0, 1), y, 1e-5, 3e-6)
    for k, want in (("dx", x.grad.transpose(0, 1)), ("dw_ih", w_ih.grad), ("dw_hh", w_hh.grad), ("db", b_ih.grad),
                    ("dh0", h0.grad), ("dc0", c0.grad)):
        _close(g[k], want, 5e-4, 5e-5)
    assert ops.lstm_bwd_caps(DEV, H, H) != 0          # back on the default arithmetic: the fp16 step path again


def test_lstm_stepper_refuses_other_shapes(ops):
    with pytest.raises(RuntimeError, match="not supported"):
        ops.LstmStepper(16, 6, 128, DEV)
    with pytest.raises(RuntimeError, match="not supported"):
        ops.LstmStepper(16, 300, 256, DEV)


def test_lstm_stepper_two_layers_through_piece_planes(ops):
    """Layer 2 of a stepped stack reads layer 1's h_t from its piece planes (`below`) instead of the f32 y rows; the
    sequence call converts its wide input to piece planes a chunk of steps at a time: both BIT-identical to each other
    and to the f32-input kernel (UAV_DEBUG_X_F32 is the A/B switch of the sequence driver)."""
    H, N, T, I = 256, 80, 19, 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, T, I, generator=g).to(DEV)
    keep = (torch.rand(N, T, generator=g) > 0.2).float()
    keep[:, 0] = 1.0
    keep = keep.to(DEV)
    mk = lambda *shape: (torch.randn(*shape, generator=g) * 0.1).to(DEV)
    W = [(mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)), (mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H))]
    h0, c0 = [mk(N, H), mk(N, H)], [mk(N, H), mk(N, H)]
    # sequence: layer 1 then layer 2 (chunked piece conversion of y1)
    y1, _, _, st1 = ops.lstm_fwd(x, keep, h0[0], c0[0], *W[0])
    y2, hn2, cn2, st2 = ops.lstm_fwd(y1, keep, h0[1], c0[1], *W[1])
    ops.set_debug_flags("x_f32")
    try:
        y2f, _, _, st2f = ops.lstm_fwd(y1, keep, h0[1], c0[1], *W[1])
    finally:
        ops.set_debug_flags()
    assert torch.equal(y2, y2f) and _stash_equal(st2, st2f)
    # stepped: both layers per time step, layer 2 fed from layer 1's piece planes
    sp = [ops.LstmStepper(N, I, H, DEV), ops.LstmStepper(N, H, H, DEV)]
    for l in range(2):
        sp[l].begin(*W[l], h0[l], c0[l])
    ys = [torch.zeros(N, T, H, device=DEV) for _ in range(2)]
    ss = [torch.zeros(N, T, 6 * H, device=DEV) for _ in range(2)]
    for t in range(T):
        kt = keep[:, t].contiguous()
        sp[0].step(x, t, ys[0], ss[0], keep=kt)
        sp[1].step(ys[0], t, ys[1], ss[1], below=sp[0], keep=kt)
    assert torch.equal(ys[0], y1) and _stash_equal(ss[0], st1)
    assert torch.equal(ys[1], y2) and _stash_equal(ss[1], st2)
    assert torch.equal(sp[1].hn, hn2) and torch.equal(sp[1].cn, cn2)


def test_lstm_bwd_stack_three_layers_equals_per_layer_calls(ops):
    """uav_lstm_bwd_stack pipelines the layers' BPTTs on internal streams (layer l + 1 from the top starts step t when layer
    l has produced dx[:, t]); three layers exercise the whole wave front.  Gate gradients and input gradients are
    BIT-identical to one uav_lstm_bwd per layer, top down."""
    H, N, T = 256, 48, 9
    g = torch.Generator().manual_seed(11)
    mk = lambda *shape: (torch.randn(*shape, generator=g) * 0.1).to(DEV)
    keep = (torch.rand(N, T, generator=g) > 0.2).float()
    keep[:, 0] = 1.0
    keep = keep.to(DEV)
    x = mk(N, T, H)
    W = [(mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H)) for _ in range(3)]
    stashes, ys = [], []
    for l in range(3):                                   # bottom -> top forward
        y, _, _, st = ops.lstm_fwd(x if l == 0 else ys[-1], keep, mk(N, H), mk(N, H), *W[l])
        ys.append(y); stashes.append(st)
    dy = (torch.randn(N, T, H, generator=g) * 1e-4).to(DEV)
    # reference: one call per layer, top down, dx fused into the backward
    assert ops.lstm_bwd_caps(DEV, H, H) & 4
    ref_dg, ref_dx, cur = {}, {}, dy
    for l in (2, 1, 0):
        r = ops.lstm_bwd(x if l == 0 else ys[l - 1], keep, stashes[l], W[l][0], W[l][1], ys[l], torch.zeros(N, H, device=DEV), dy=cur,
                         need_dx=(l > 0), want_dstate=False)
        ref_dg[l], ref_dx[l] = r["dgates"].clone(), None if r["dx"] is None else r["dx"].clone()
        cur = r["dx"]
    specs = [{"stash": stashes[l], "w_hh": W[l][1], "w_ih": W[l][0] if l > 0 else None,
              "dgates": ops.lstm_dgates(N, T, H, DEV), "dx": torch.zeros(N, T, H, device=DEV) if l > 0 else None} for l in (2, 1, 0)]
    ops.lstm_bwd_stack(specs, keep, dy=dy)
    torch.cuda.synchronize()
    for s, l in zip(specs, (2, 1, 0)):
        assert torch.equal(s["dgates"], ref_dg[l]), l
        if l > 0:
            assert torch.equal(s["dx"], ref_dx[l]), l


@pytest.mark.parametrize("N,T,I", [(80, 19, 8), (33, 9, 256), (64, 12, 256), (5, 6, 6), (130, 7, 3), (16, 1, 8), (70, 2, 256), (256, 33, 256)])
def test_h256_gate_gradients_stored_once_equal_the_f32_rows_form(ops, N, T, I):
    """h = 256 on the fp16-split arithmetic keeps the gate gradients ONCE, as the fp16 piece chunks the BPTT's recurrent product
    consumes (+ one power-of-two scale per env and step; common.h DgPack, uav_lstm_dgates_bytes), and the weight-gradient pass
    reads that form directly (csrc/wgrad_pc.hip: LDS-DMA + transposed fragment reads, the scales riding on h_prev / x).  Against the
    round-4 form kept behind UAV_DEBUG_DG_F32 (f32 rows written as well, weight gradients through gemm_h3_tn8_kernel):
    the recurrent results are the same bits (same pieces, same kernel), the unpacked gate gradients equal the f32 rows to the
    pieces' 2^-22 of each row's largest magnitude, the weight gradients agree to f32 round-off of their largest element --
    ragged env counts (rows past N are zero pieces with scale 0), narrow and hidden-wide inputs, restart masks."""
    H = 256
    g = torch.Generator().manual_seed(N * 7 + T)
    mk = lambda *shape, s=0.1: (torch.randn(*shape, generator=g) * s).to(DEV)
    keep = (torch.rand(N, T, generator=g) > 0.2).float().to(DEV)
    x = mk(N, T, I, s=1.0)
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)
    h0, c0 = mk(N, H, s=0.3), mk(N, H, s=0.3)
    # gradients spanning many binades across envs: the per-(env, step) scales are what keeps the small rows' bits
    dy = mk(N, T, H, s=1.0) * torch.logspace(-9, 0, N, device=DEV).view(N, 1, 1)
    dheads = mk(N, T, 6, s=1.0)
    need_dx = I == H
    out = {}
    for mode in ("packed", "rows"):
        ops.set_debug_flags(*(("dg_f32",) if mode == "rows" else ()))
        # (the forward pass belongs to the mode too: only the f32-rows form writes -- and reads -- the stash's h_prev slot)
        y, _, _, stash = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh)
        assert (ops.lstm_dgates_bytes(N, T, H, DEV) == N * T * 4 * H * 4) == (mode == "rows")
        r = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dy=dy, dhn=mk(N, H) * 0, dcn=None, need_dx=need_dx, wgrad_dheads=dheads)
        r["rows"] = ops.lstm_dgates_f32(r["dgates"], N, T, H)
        torch.cuda.synchronize()
        out[mode] = r
    # a buffer written in one form is refused by a weight-gradient call that would read the other (mode changed in between)
    with pytest.raises(RuntimeError, match="must not change between"):
        ops.lstm_wgrad(x, keep, h0, y, stash, out["packed"]["dgates"], w_ih)       # flags still say f32 rows
    ops.set_debug_flags()
    a, b = out["packed"], out["rows"]
    for k in ("dh0", "dc0") + (("dx",) if need_dx else ()):
        assert torch.equal(a[k], b[k]), k
    rowmax = b["rows"].abs().amax(dim=2, keepdim=True)
    assert ((a["rows"] - b["rows"]).abs() <= rowmax * 2.0 ** -21 + 1e-37).all()
    assert torch.equal(a["dw_head"], b["dw_head"])
    for k in ("db", "dw_hh", "dw_ih"):
        want = b[k].double().cpu()
        err = (a[k].double().cpu() - want).abs().max().item()
        assert err <= 2e-6 * want.abs().max().item(), (k, err, want.abs().max().item())


def test_lstm_bwd_stack_rejects_a_bad_layer_before_queueing_work(ops):
    """A layer that feeds the one below without w_ih / dx is an argument error: uav_lstm_bwd_stack must return it BEFORE forking
    its side streams (all layers are validated first), leave nothing queued, and the next valid call must give the same bits
    as a fresh one.  Also: the debug-flag setter refuses unknown bits."""
    H, N, T = 256, 32, 5
    g = torch.Generator().manual_seed(3)
    mk = lambda *shape: (torch.randn(*shape, generator=g) * 0.1).to(DEV)
    x = mk(N, T, H)
    W = [(mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H)) for _ in range(2)]
    y0, _, _, st0 = ops.lstm_fwd(x, None, mk(N, H), mk(N, H), *W[0])
    y1, _, _, st1 = ops.lstm_fwd(y0, None, mk(N, H), mk(N, H), *W[1])
    dy = (torch.randn(N, T, H, generator=g) * 1e-4).to(DEV)

    def specs(bad):
        return [{"stash": st1, "w_hh": W[1][1], "w_ih": None if bad else W[1][0], "dgates": ops.lstm_dgates(N, T, H, DEV),
                 "dx": None if bad else torch.zeros(N, T, H, device=DEV)},
                {"stash": st0, "w_hh": W[0][1], "w_ih": None, "dgates": ops.lstm_dgates(N, T, H, DEV), "dx": None}]
    good = specs(False)
    ops.lstm_bwd_stack(good, None, dy=dy)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="w_ih and dx are required"):
        ops.lstm_bwd_stack(specs(True), None, dy=dy)
    again = specs(False)
    ops.lstm_bwd_stack(again, None, dy=dy)
    torch.cuda.synchronize()
    for a, b in zip(good, again):
        assert torch.equal(a["dgates"], b["dgates"])
    from uavppo import _lib
    assert _lib.lib().uav_set_debug_flags(ops.Context.get(DEV).handle, 64) != 0
    ops.set_debug_flags()
