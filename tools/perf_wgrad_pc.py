#!/usr/bin/env python3
"""uav_lstm_wgrad at C5's layer shapes (4096 envs x 256 steps, h = 256; input 8 = layer 1, 256 = layer 2) on whatever build UAVPPO_LIB
names, from gate gradients produced by a real uav_lstm_bwd: ms per call.  Flags: debug-flag names (dg_f32 = the round-4 form)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402

dev = "cuda:0"
N, T, H = int(os.environ.get("PC_N", 4096)), int(os.environ.get("PC_T", 256)), 256
flags = tuple(a for a in sys.argv[1:])
ops.set_debug_flags(*flags)
g = torch.Generator().manual_seed(1)
mk = lambda *s, sc=0.05: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
for I in (8, 256):
    x = mk(N, T, I, sc=1.0)
    w_ih, w_hh, b = mk(4 * H, I), mk(4 * H, H), mk(4 * H)
    h0 = mk(N, H)
    y, _, _, stash = ops.lstm_fwd(x, None, h0, h0, w_ih, w_hh, b, b)
    dy = mk(N, T, H, sc=1e-3)
    r = ops.lstm_bwd(x, None, stash, w_ih, w_hh, y, h0, dy=dy, need_dx=(I == H), want_dstate=False)
    dg = r["dgates"]
    fn = lambda: ops.lstm_wgrad(x, None, h0, y, stash, dg, w_ih)  # noqa: E731
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 3)
    print(f"{os.path.basename(os.environ.get('UAVPPO_LIB', 'in-tree')):26s} {' '.join(flags):8s} I = {I:3d}: uav_lstm_wgrad {float(np.median(ts)):.3f} ms", flush=True)
    del x, y, stash, dy, r, dg
