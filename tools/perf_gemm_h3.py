#!/usr/bin/env python3
"""uav_gemm_f16x3 against uav_gemm_f32 at the C5 weight-gradient / input-gradient shapes (1 M rows, 1024 gate rows, 256)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402


def timeit(fn, n=5, burst=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(burst):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / burst)
    return float(np.median(ts))


def main():
    dev = "cuda:0"
    NT = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    dg = torch.randn(NT, 1024, device=dev) * 1e-6
    stash = torch.rand(NT, 6 * 256, device=dev) * 2 - 1
    hprev = stash[:, 5 * 256:]                       # strided view like the BPTT stash
    y1 = torch.rand(NT, 256, device=dev) * 2 - 1
    w = torch.randn(1024, 256, device=dev) * 0.05
    amax = ops.absmax(dg)
    flop = 2.0 * NT * 1024 * 256
    for name, call in (
        ("dW = dG^T y   (TN, K = rows)", lambda h3: ops.gemm(dg, y1, trans_a=True, split_fp16=h3, a_absmax=amax if h3 else None)),
        ("dx = dG W_ih  (NN, M = rows)", lambda h3: ops.gemm(dg, w, split_fp16=h3, a_absmax=amax if h3 else None)),
    ):
        t32 = timeit(lambda: call(False))
        t16 = timeit(lambda: call(True))
        a, b = call(False), call(True)
        rel = ((a - b).abs().max() / a.abs().max()).item()
        print(f"{name}: exact-f32 {t32:.3f} ms ({flop / t32 / 1e9:.0f} TF)   split-fp16 {t16:.3f} ms ({flop / t16 / 1e9:.0f} TF f32-equiv, "
              f"{3 * flop / t16 / 1e9:.0f} TF executed)   max |diff| / max = {rel:.2e}", flush=True)
    # the dW shape on the two split-fp16 kernels: the TN kernel (default) against the general one (UAV_DEBUG_GEMM_TN_OFF); same bits
    outs = {}
    for name, flags in (("tn kernel", ()), ("older kernel", ("gemm_tn_off",))):
        ops.set_debug_flags(*flags)
        for bname, b in (("y [rows x 256]", y1),):
            t = timeit(lambda: ops.gemm(dg, b, trans_a=True, split_fp16=True, a_absmax=amax))
            outs[(name, bname)] = ops.gemm(dg, b, trans_a=True, split_fp16=True, a_absmax=amax)
            print(f"dW = dG^T {bname:30s} {name:13s} {t:.3f} ms ({3 * flop / t / 1e9:.0f} TF executed, "
                  f"{(dg.numel() + b.numel()) * 4 / t / 1e6:.0f} GB/s of operand bytes)", flush=True)
    ops.set_debug_flags()
    for bname in ("y [rows x 256]",):
        same = torch.equal(outs[("tn kernel", bname)], outs[("older kernel", bname)])
        print(f"  {bname}: tn kernel == older kernel bit for bit: {same}")
    t = timeit(lambda: ops.colsum(dg))
    print(f"colsum [rows x 1024]: {t:.3f} ms ({dg.numel() * 4 / t / 1e6:.0f} GB/s)")


if __name__ == "__main__":
    main()
