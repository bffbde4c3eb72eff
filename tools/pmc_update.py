#!/usr/bin/env python3
"""Small driver for PMC collection: a few update iterations at C3 (no rollout timing)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402
tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", use_curriculum=False, epochs=2)
for _ in range(2):
    tr.collect()
    tr.update()
torch.cuda.synchronize()
