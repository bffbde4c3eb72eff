#!/usr/bin/env python3
"""Measured HBM ceilings on this MI355X ('confirm with a device-copy microbench and quote the measured ceiling'):
a device-to-device copy (read + write), a fill (write only) and a column-sum read stream (uav_colsum, read only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402


def timeit(fn, n=7, burst=4):
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
    return float(np.median(ts)) * 1e-3


def main():
    dev = "cuda:0"
    n = 1 << 30                                         # 4 GiB of f32
    src, dst = torch.empty(n, device=dev).normal_(), torch.empty(n, device=dev)
    t = timeit(lambda: dst.copy_(src))
    print(f"copy  4 GiB -> 4 GiB : {2 * 4 * n / t / 1e12:.2f} TB/s (read + write)")
    t = timeit(lambda: dst.fill_(1.0))
    print(f"fill  4 GiB          : {4 * n / t / 1e12:.2f} TB/s (write only)")
    x = src.view(n // 1024, 1024)
    t = timeit(lambda: ops.colsum(x))
    print(f"colsum [1M x 1024]   : {4 * n / t / 1e12:.2f} TB/s (read only, uav_colsum)")


if __name__ == "__main__":
    main()
