#!/usr/bin/env python3
"""Small driver for PMC collection with the reference's own MLP policy at BASELINE C3's per-GPU shape (4096 envs x 128 steps, fused
kernels of csrc/mlp_fused.hip): ONE iteration, 2 epochs (counters are per launch)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402
tr = VecPPOTrainer(4096, 128, "mlp", device="cuda:0", use_curriculum=False, epochs=2)
tr.collect()
tr.update()
torch.cuda.synchronize()
