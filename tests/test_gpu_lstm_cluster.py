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
