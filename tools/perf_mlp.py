#!/usr/bin/env python3
"""MLP (the reference's PPOActorCritic) path at the C3 buffer shape: step-wise rollout + fused update."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch
from uavppo.trainer import VecPPOTrainer
N, T = 4096, 128
tr = VecPPOTrainer(N, T, "mlp", device="cuda:0", use_curriculum=False)
for name, fn in (("collect (step-wise, 128 steps)", tr.collect), ("update (GAE + 5 epochs)", tr.update)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:34s}: {dt*1e3:8.2f} ms  -> {N*T/dt/1e6:7.2f} M env-steps/s")
