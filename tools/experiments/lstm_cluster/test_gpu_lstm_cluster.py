"""h = 256 persistent cluster kernels (csrc/lstm_cluster.hip: weights resident in registers, 8 workgroups per 64-env tile
exchanging h_t through L2) against the one-launch-per-step kernels they replace: the arithmetic and its order are the same,
so every output must be BIT-identical -- stash rows, layer output, final state -- for ragged env counts, restart masks,
several tiles per cluster (both step parities across a tile boundary) and both input widths (a first layer reading
observations, a stacked layer reading the layer below).  And against torch.nn.LSTM (PPOV2.0/model.py:206-212 semantics).
No bounded wait may have run out.  -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = 256


def _case(N, T, I, seed, mask_p=0.15):
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(N, T, I, generator=g)
    keep = (torch.rand(N, T, generator=g) > mask_p).float()
    h0, c0 = torch.randn(N, H, generator=g) * 0.5, torch.randn(N, H, generator=g) * 0.5
    k = 1.0 / H ** 0.5
    w_ih, w_hh = (torch.rand(4 * H, I, generator=g) * 2 - 1) * k, (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
    b_ih, b_hh = (torch.rand(4 * H, generator=g) * 2 - 1) * k, (torch.rand(4 * H, generator=g) * 2 - 1) * k
    return [t.to(DEV).contiguous() for t in (x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh)]


def _both(args, *extra):
    from uavppo import ops
    ops.set_debug_flags()
    ref = ops.lstm_fwd(*args)
    torch.cuda.synchronize()
    ops.set_debug_flags("cluster", *extra)
    try:
        got = ops.lstm_fwd(*args)
        torch.cuda.synchronize()
    finally:
        ops.set_debug_flags()
    return ref, got


@pytest.mark.parametrize("N,T,I", [(64, 6, 8), (100, 5, 8), (37, 9, 6), (200, 12, 8), (64, 4, 256), (150, 7, 256),
                                   (2048 + 64 * 3 + 5, 4, 8), (4096, 3, 8), (4096, 2, 256), (4096 + 64, 5, 256)])
def test_cluster_forward_is_bit_identical_to_the_per_step_path(N, T, I):
    from uavppo import ops
    args = _case(N, T, I, seed=N * 31 + T * 7 + I)
    e0 = ops.lstm_cluster_errors()
    (y0, hn0, cn0, st0), (y1, hn1, cn1, st1) = _both(args)
    assert ops.lstm_cluster_errors() == e0, "a bounded wait of the cluster kernel ran out"
    assert torch.isfinite(y1).all()
    assert torch.equal(y0, y1) and torch.equal(hn0, hn1) and torch.equal(cn0, cn1)
    assert torch.equal(st0, st1)


@pytest.fixture
def cluster_on():
    from uavppo import ops
    ops.set_debug_flags("cluster")
    yield
    ops.set_debug_flags()


@pytest.mark.parametrize("N,T,I", [(4096, 9, 8), (4096 + 64, 6, 256), (333, 12, 8)])
def test_cluster_forward_hand_off_by_write_through_stores(N, T, I):
    """The hand-off payload normally stays in the L2 of the XCD a cluster was MEASURED to sit on (plain stores, L1-bypassing
    loads); clusters spread over several XCDs use write-through (sc1) stores.  Force that form: same bits."""
    from uavppo import ops
    args = _case(N, T, I, seed=N + T + I)
    (y0, hn0, cn0, st0), (y1, hn1, cn1, st1) = _both(args, "cluster_sc1")
    assert ops.lstm_cluster_errors() == 0
    assert torch.equal(y0, y1) and torch.equal(hn0, hn1) and torch.equal(cn0, cn1) and torch.equal(st0, st1)


def test_cluster_forward_matches_torch_lstm(cluster_on):
    """Two stacked layers (I = 8 -> 256 -> 256) on the cluster kernels against torch.nn.LSTM on the CPU, restart masks applied
    the reference way (state zeroed where an episode ended)."""
    from uavppo import ops
    N, T = 96, 10
    a1 = _case(N, T, 8, seed=5, mask_p=0.1)
    x, keep, h0, c0 = a1[:4]
    a2 = _case(N, T, 256, seed=6)
    y1, *_ = ops.lstm_fwd(*a1)
    y2, hn2, cn2, _ = ops.lstm_fwd(y1, keep, a2[2], a2[3], *a2[4:])
    lstm = torch.nn.LSTM(8, H, num_layers=2, batch_first=True)
    with torch.no_grad():
        for l, a in enumerate((a1, a2)):
            getattr(lstm, f"weight_ih_l{l}").copy_(a[4].cpu()); getattr(lstm, f"weight_hh_l{l}").copy_(a[5].cpu())
            getattr(lstm, f"bias_ih_l{l}").copy_(a[6].cpu()); getattr(lstm, f"bias_hh_l{l}").copy_(a[7].cpu())
        h = torch.stack([a1[2].cpu(), a2[2].cpu()]); c = torch.stack([a1[3].cpu(), a2[3].cpu()])
        outs = []
        for t in range(T):
            k = keep[:, t].cpu()[None, :, None]
            o, (h, c) = lstm(x[:, t:t + 1].cpu(), (h * k, c * k))
            outs.append(o)
        want = torch.cat(outs, 1)
    assert torch.allclose(y2.cpu(), want, atol=3e-6, rtol=1e-5)
    assert torch.allclose(hn2.cpu(), h[1], atol=3e-6) and torch.allclose(cn2.cpu(), c[1], atol=5e-6)


def test_cluster_forward_is_deterministic_and_tiles_are_independent(cluster_on):
    from uavppo import ops
    args = _case(4096, 6, 8, seed=77)
    y, hn, cn, st = ops.lstm_fwd(*args)
    y2, hn2, cn2, st2 = ops.lstm_fwd(*args)
    assert torch.equal(y, y2) and torch.equal(st, st2)
    sl = slice(64 * 40, 64 * 41)                    # one tile alone (handled by another cluster, first instead of second tile)
    sub = [a[sl].contiguous() for a in args[:4]] + args[4:]
    y3, hn3, cn3, st3 = ops.lstm_fwd(*sub)
    assert torch.equal(y[sl], y3) and torch.equal(st[sl], st3) and torch.equal(cn[sl], cn3)
    assert ops.lstm_cluster_errors() == 0


# ------------------------------------------------------------------------------------------------ backward
def _bwd_case(N, T, I, seed, top):
    """A layer's forward (per-step path) and an upstream gradient: dy [N,T,H], or for the top layer dheads [N,T,6] + w_head."""
    from uavppo import ops
    args = _case(N, T, I, seed, mask_p=0.1)
    ops.set_debug_flags()
    y, hn, cn, stash = ops.lstm_fwd(*args)
    g = torch.Generator("cpu").manual_seed(seed + 1)
    up = {}
    if top:
        up["dheads"] = (torch.randn(N, T, 6, generator=g) / (N * T)).to(DEV)
        up["w_head"] = (torch.randn(6, H, generator=g) * 0.1).to(DEV)
    else:
        up["dy"] = (torch.randn(N, T, H, generator=g) / (N * T)).to(DEV)
    return args, y, stash, up


def _run_bwd(args, y, stash, up, need_dx, *flags, arith="fp16x3"):
    from uavppo import ops
    x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh = args
    ops.set_debug_flags(*flags)
    ops.set_lstm_arith(arith)
    try:
        r = ops.lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, need_dx=need_dx, **up)
        torch.cuda.synchronize()
    finally:
        ops.set_debug_flags()
        ops.set_lstm_arith("fp16x3")
    return {k: r[k].clone() for k in ("dgates", "dh0", "dc0", "dx", "dw_hh", "db") if r.get(k) is not None}


@pytest.mark.parametrize("N,T,I,top", [(64, 5, 8, False), (100, 7, 8, True), (64, 6, 256, False), (150, 5, 256, True),
                                       (4096, 4, 8, False), (4096 + 64, 3, 256, True), (2048 + 37, 5, 256, False)])
def test_cluster_backward_has_the_per_step_paths_accuracy(N, T, I, top):
    """The cluster BPTT scales an env's gate gradients per WORKGROUP (128 gate rows) instead of per env (1024), so its products
    round differently from the per-step kernels': judged against the exact-f32-MFMA path, its error must stay within twice
    theirs (+ f32 noise).  dgates of the LAST step involve no product: bit-identical.  No bounded wait may have run out."""
    from uavppo import ops
    args, y, stash, up = _bwd_case(N, T, I, seed=N + 3 * T + I, top=top)
    need_dx = I == 256
    up_ref = up if not top else {"dy": (up["dheads"].reshape(-1, 6) @ up["w_head"]).reshape(N, T, H)}      # (that path takes dy only)
    ref = _run_bwd(args, y, stash, up_ref, need_dx, arith="f32_mfma")
    step = _run_bwd(args, y, stash, up, need_dx)
    e0 = ops.lstm_cluster_errors()
    clu = _run_bwd(args, y, stash, up, need_dx, "cluster")
    assert ops.lstm_cluster_errors() == e0
    assert torch.equal(clu["dgates"][:, T - 1], step["dgates"][:, T - 1])
    for k in ref:
        assert torch.isfinite(clu[k]).all(), k
        scale = ref[k].abs().max().item()
        es, ec = (step[k] - ref[k]).abs().max().item(), (clu[k] - ref[k]).abs().max().item()
        assert ec <= 2.0 * es + 2e-6 * scale, (k, ec, es, scale)


def test_cluster_backward_stack_equals_layer_by_layer_and_is_deterministic():
    """uav_lstm_bwd_stack on the cluster kernels (top layer from dheads, dx handed down) == one uav_lstm_bwd per layer, bit for
    bit, twice."""
    from uavppo import ops
    N, T = 300, 6
    a1, y1, st1, _ = _bwd_case(N, T, 8, seed=41, top=False)
    a2 = _case(N, T, 256, seed=42, mask_p=0.1)
    a2[0], a2[1] = y1, a1[1]
    ops.set_debug_flags()
    y2, _, _, st2 = ops.lstm_fwd(*a2)
    g = torch.Generator("cpu").manual_seed(7)
    dheads = (torch.randn(N, T, 6, generator=g) / (N * T)).to(DEV)
    w_head = (torch.randn(6, H, generator=g) * 0.1).to(DEV)
    keep = a1[1]

    def stack():
        layers = [dict(stash=st2, w_hh=a2[5], w_ih=a2[4], dgates=torch.empty(N, T, 4 * H, device=DEV), dx=torch.empty(N, T, H, device=DEV)),
                  dict(stash=st1, w_hh=a1[5], w_ih=None, dgates=torch.empty(N, T, 4 * H, device=DEV), dx=None)]
        ops.lstm_bwd_stack(layers, keep, dheads=dheads, w_head=w_head)
        torch.cuda.synchronize()
        return layers
    ops.set_debug_flags("cluster")
    try:
        la, lb = stack(), stack()
        top = ops.lstm_bwd(a2[0], keep, st2, a2[4], a2[5], y2, a2[2], dheads=dheads, w_head=w_head, need_dx=True)
        low = ops.lstm_bwd(a1[0], keep, st1, a1[4], a1[5], y1, a1[2], dy=top["dx"])
        torch.cuda.synchronize()
    finally:
        ops.set_debug_flags()
    for l in range(2):
        assert torch.equal(la[l]["dgates"], lb[l]["dgates"])
    assert torch.equal(la[0]["dx"], lb[0]["dx"])
    assert torch.equal(la[0]["dgates"], top["dgates"]) and torch.equal(la[0]["dx"], top["dx"])
    assert torch.equal(la[1]["dgates"], low["dgates"])
    assert ops.lstm_cluster_errors() == 0
