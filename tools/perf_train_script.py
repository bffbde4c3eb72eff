#!/usr/bin/env python3
"""How fast is the reference-shaped entry point itself?  train_ppo2.0.py's train_ppo_vectorised at BASELINE C3's shape (4096 envs x 128
steps, LSTM h=128), with its per-episode CSV rows -- against bench.py's rate for the bare trainer loop.  usage: perf_train_script.py [iterations=60]"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")
sys.path[:0] = [ROOT, PKG]
import config  # noqa: E402
import torch  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for k, v in dict(NUM_ENVS=4096, HORIZON=128, POLICY="lstm", HIDDEN=128, NUM_LAYERS=1).items():
    setattr(config, k, v)
spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(PKG, "train_ppo2.0.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
m.train_ppo_vectorised(iterations=5, csv_path=None, model_path=None, log_every=0)          # warm-up (allocations)
torch.cuda.synchronize()


def run(n):
    t0 = time.perf_counter()
    tr, rows = m.train_ppo_vectorised(iterations=n, csv_path="/tmp/perf_train_script.csv", model_path=None, log_every=0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, len(rows)


t_small, _ = run(10)
t_big, nrows = run(10 + iters)
dt = t_big - t_small                     # the trainer's construction and the CSV write of the common part cancel
print(f"train_ppo_vectorised at C3's shape: {iters} iterations of 4096 x 128 in {dt:.3f} s = {iters * 4096 * 128 / dt / 1e6:.1f} M env-steps/s "
      f"({1e3 * dt / iters:.2f} ms per iteration; {nrows} CSV rows in the long run; difference of a {10 + iters}- and a 10-iteration call)")
