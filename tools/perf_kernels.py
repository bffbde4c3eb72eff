#!/usr/bin/env python3
"""A/B timing of the hot kernels on one GPU (HIP events, interleaved rounds in one process)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    N, T, H = 4096, 128, 128
    dev = "cuda:0"
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, device=dev, use_curriculum=False)
    print("rollout procedural          : %.3f ms (min %.3f)" % timeit(lambda: tr.collect()))
    g = torch.Generator(device=dev).manual_seed(0)
    fa = torch.randint(0, 5, (N, T), generator=g, device=dev, dtype=torch.int32)
    nz = torch.randn(N, T, 2, generator=g, device=dev, dtype=torch.float64)
    print("rollout forced act + noise  : %.3f ms (min %.3f)" % timeit(lambda: tr.collect(forced_act=fa, noise=nz)))
    # materialised bank (64 fields, synthetic)
    F = 64
    bank = torch.rand(F, 500, 500, 2, device=dev, dtype=torch.float64) * 9
    src = torch.rand(F, 2, device=dev, dtype=torch.float64) * 400 + 50
    tb = VecPPOTrainer(N, T, "lstm", hidden=H, device=dev, use_curriculum=False, bank=bank, bank_sources=src)
    print("rollout bank                : %.3f ms (min %.3f)" % timeit(lambda: tb.collect()))
    print("rollout bank + forced+noise : %.3f ms (min %.3f)" % timeit(lambda: tb.collect(forced_act=fa, noise=nz)))
    tr.collect()
    tr.compute_advantages()
    pol = tr.policy
    b = tr.buf

    def fwd():
        return pol.heads(b["obs"], b["keep"], tr.h0, tr.c0, tr.work)
    print("lstm fwd + heads            : %.3f ms (min %.3f)" % timeit(fwd))
    v = pol.views
    print("lstm fwd kernel (stash)     : %.3f ms (min %.3f)" % timeit(lambda: ops.lstm_fwd(
        b["obs"], b["keep"], tr.h0[0], tr.c0[0], v["lstm.weight_ih_l0"], v["lstm.weight_hh_l0"], v["lstm.bias_ih_l0"],
        v["lstm.bias_hh_l0"], stash=tr.work["stash0"], y=tr.work["y0"])))
    print("lstm fwd kernel (no stash)  : %.3f ms (min %.3f)" % timeit(lambda: ops.lstm_fwd(
        b["obs"], b["keep"], tr.h0[0], tr.c0[0], v["lstm.weight_ih_l0"], v["lstm.weight_hh_l0"], v["lstm.bias_ih_l0"],
        v["lstm.bias_hh_l0"], want_stash=False, y=tr.work["y0"])))
    heads = fwd()
    dheads = torch.randn_like(heads) / heads.shape[0]
    ops.KERNEL_TIMER.enable(("lstm_bwd", "lstm_wgrad"))
    for _ in range(10):
        fwd()
        pol.backward(dheads, tr.work, tr.dhead_bias)
    for k, s in ops.KERNEL_TIMER.summary().items():
        print("%-28s: %.3f ms avg over %d" % (k, s["avg_ms"], s["n"]))


if __name__ == "__main__":
    main()
