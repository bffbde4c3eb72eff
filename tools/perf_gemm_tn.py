#!/usr/bin/env python3
"""The dW-shaped split-fp16 product (1 M rows x 1024 gate rows x 256) on whatever build UAVPPO_LIB names: ms and rates."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402

dev = "cuda:0"
NT = 1 << 20
dg = torch.randn(NT, 1024, device=dev) * 1e-6
y1 = torch.rand(NT, 256, device=dev) * 2 - 1
amax = ops.absmax(dg)
flags = tuple(a for a in sys.argv[1:])
ops.set_debug_flags(*flags)
fn = lambda: ops.gemm(dg, y1, trans_a=True, split_fp16=True, a_absmax=amax)  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 5)
t = float(np.median(ts))
flop = 2.0 * NT * 1024 * 256
print(f"{os.path.basename(os.environ.get('UAVPPO_LIB', 'in-tree')):28s} {' '.join(flags):12s} {t:.3f} ms  {3 * flop / t / 1e9:.0f} TF executed  "
      f"{(dg.numel() + y1.numel()) * 4 / t / 1e6:.0f} GB/s operand bytes")
