#!/usr/bin/env python3
"""Does the fused path learn?  A few hundred PPO iterations at a C3-like shape (or, with a 4th argument "c5", the stacked
h = 256 x 2 + trend policy of BASELINE C5 at 1024 envs x 128 steps: stepper rollout, piece-plane inputs, fused dx, split
GEMMs); prints reward / success trend."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo.trainer import VecPPOTrainer  # noqa: E402


def main():
    N, T = 4096, 128
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
    mb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    if len(sys.argv) > 4 and sys.argv[4] in ("c5", "c5full"):
        N, T = (1024, 128) if sys.argv[4] == "c5" else (4096, 256)       # c5full: BASELINE C5's exact per-GPU shape
        tr = VecPPOTrainer(N, T, "lstm", hidden=256, layers=2, trend_k=2, variant="v2.1", device="cuda:0", seed=1, lr=lr,
                           num_minibatches=mb, use_curriculum=True)
    elif len(sys.argv) > 4 and sys.argv[4] == "mlp":      # the reference's own policy through the fused MLP kernels
        tr = VecPPOTrainer(N, T, "mlp", device="cuda:0", seed=1, lr=lr, num_minibatches=mb, use_curriculum=True)
    else:
        tr = VecPPOTrainer(N, T, "lstm", hidden=128, device="cuda:0", seed=1, lr=lr, num_minibatches=mb, use_curriculum=True)
    t0 = time.perf_counter()
    for it in range(iters):
        tr.train_iteration()
        if (it + 1) % 25 == 0:
            fl = tr.buf["flags"]
            ended = (fl & 1).sum().item()
            reached = ((fl >> 1) & 1).sum().item()
            pl, vl, ent = tr.losses()
            print(f"it {it + 1:4d}  mean reward/step {tr.buf['rew'].mean().item():8.4f}  episodes ended {int(ended):6d}  "
                  f"reached {int(reached):6d} ({100.0 * reached / max(ended, 1):5.1f}%)  radius {tr.radius:5.1f}  entropy {ent:.3f}  "
                  f"value loss {vl:.3f}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{iters} iterations, {iters * N * T / 1e6:.0f} M env-steps in {dt:.1f} s = {iters * N * T / dt / 1e6:.1f} M env-steps/s")


if __name__ == "__main__":
    main()
