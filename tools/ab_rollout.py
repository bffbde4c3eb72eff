#!/usr/bin/env python3
"""Same-box A/B of the fused rollouts: median HIP-event time of one collect() at the C3 shape, LSTM and MLP policy.
    UAVPPO_LIB=<build> python tools/ab_rollout.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

for kind in ("lstm", "mlp"):
    tr = VecPPOTrainer(4096, 128, kind, hidden=128, device="cuda:0", seed=1, use_curriculum=False)
    for _ in range(5):
        tr.collect()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        tr.collect()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    print("%s %-4s rollout: median %.1f us, min %.1f us" % (os.path.basename(os.environ.get("UAVPPO_LIB", "in-tree")), kind,
                                                          1e3 * float(np.median(ts)), 1e3 * min(ts)))
