#!/usr/bin/env python3
"""A/B of the iteration loop on one device, interleaved rounds (cdna_hip_programming.md rule 24):
  old  range guard off          new  range guard on (uavppo/trainer.py: Adam publishes max |param|, polled at the curriculum sync)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", seed=1234)


def run(mode, k=10):
    if mode == "old":
        tr._guarded = lambda: False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        tr.train_iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k * 1e3
    if mode == "old":
        del tr._guarded
    return dt


for m in ("old", "new"):
    run(m, 3)
res = {"old": [], "new": []}
for r in range(6):
    for m in ("old", "new"):
        res[m].append(run(m))
for m, v in res.items():
    print(m, "ms/iter: median %.3f min %.3f" % (sorted(v)[len(v) // 2], min(v)), ["%.3f" % x for x in v])
