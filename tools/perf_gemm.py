#!/usr/bin/env python3
"""uav_gemm_f32 timing at the shapes of the stacked h=256 path and the MLP policy (UAV_GEMM_F32=1: exact-f32 kernels)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402


def timeit(fn, n=5, burst=20):
    for _ in range(burst):
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
    mode = "f32" if os.environ.get("UAV_GEMM_F32") else "x6"
    for (M, N, K, tb) in ((4096, 1024, 256, True), (4096, 256, 1024, False), (524288, 128, 256, True), (524288, 256, 128, False),
                          (1024, 256, 65536, None)):
        if tb is None:      # A^T B with huge K (weight gradient): a [K, M], b [K, N]
            a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            fn = lambda: ops.gemm(a, b, trans_a=True)
            ref = (a[:4096].double().T @ b[:4096].double())
            got = ops.gemm(a[:4096].contiguous(), b[:4096].contiguous(), trans_a=True)
        else:
            a = torch.randn(M, K, device=dev)
            b = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
            fn = lambda: ops.gemm(a, b, trans_b=tb)
            ref = a[:256].double() @ (b.double().T if tb else b.double())
            got = fn()[:256]
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        ms = timeit(fn)
        print(f"[{mode}] M={M} N={N} K={K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s (f32-equivalent)  rel err {err:.2e}")


if __name__ == "__main__":
    main()
