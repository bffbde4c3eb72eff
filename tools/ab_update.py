#!/usr/bin/env python3
"""Same-box A/B of the C3 sequence kernels: UAVPPO_LIB=<build> python tools/ab_update.py -> fwd / bwd / wgrad / rollout ms."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo import ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", use_curriculum=False, seed=3)
for _ in range(3):
    tr.train_iteration()
ops.KERNEL_TIMER.enable(("lstm_fwd", "lstm_bwd", "lstm_wgrad", "rollout"))
for _ in range(12):
    tr.train_iteration()
torch.cuda.synchronize()
s = ops.KERNEL_TIMER.summary()
print(os.path.basename(os.environ.get("UAVPPO_LIB", "in-tree")), " ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in sorted(s.items())))
