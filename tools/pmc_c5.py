#!/usr/bin/env python3
"""Small driver for PMC collection at BASELINE C5's per-GPU shape (4096 envs, LSTM h=256 x2, obs 6+2): ONE iteration with a short
horizon (counters are per launch; the per-step kernels are the same whatever T) and 2 epochs."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402
T = int(os.environ.get("PMC_C5_T", "32"))
tr = VecPPOTrainer(4096, T, "lstm", hidden=256, layers=2, variant="v2.1", device="cuda:0", use_curriculum=False, epochs=2, trend_k=2)
tr.collect()
tr.update()
torch.cuda.synchronize()
