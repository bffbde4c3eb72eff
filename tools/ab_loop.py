#!/usr/bin/env python3
"""A/B of the iteration loop on one device, interleaved rounds (cdna_hip_programming.md rule 24):
  guard   range guard off / on (uavppo/trainer.py: Adam publishes max |param|, polled at the next rollout)
  side    curriculum success bits packed + copied on the main stream at the iteration's host sync / on a side stream right
          behind the rollout (the sync then never drains the main stream)
usage: ab_loop.py [guard|side]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", seed=1234)


WHAT = sys.argv[1] if len(sys.argv) > 1 else "side"


def run(mode, k=10):
    if WHAT == "side":
        tr.side_stream_curriculum = (mode == "new")
    elif mode == "old":
        tr._guarded = lambda: False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        tr.train_iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k * 1e3
    if WHAT == "guard" and mode == "old":
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
