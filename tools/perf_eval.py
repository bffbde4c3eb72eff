#!/usr/bin/env python3
"""Throughput of the vectorised greedy evaluation (row N2): N parallel episodes with the PPOV2.0 stop controller."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import evaluate_with_lstm as ev  # noqa: E402
from uavppo.policy import MLPActorCritic  # noqa: E402
from uavppo.vec_env import VecMethaneEnv  # noqa: E402


def main():
    dev = "cuda:0"
    for N in (1000, 4096):
        env = VecMethaneEnv(N, "v2.0", dev, seed=1)
        pol = MLPActorCritic(6, 5, device=dev, seed=2)
        pol.views["head.weight"][:5].mul_(40.0)
        pred = ev.ConcentrationThresholdPredictor(device=dev, seed=3)
        pred.fc["fc.4.bias"].fill_(60.0)
        ctl = ev.ThresholdController(pred, (0.0, 100.0), N, device=dev)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = ev.evaluate(lambda o: pol.heads(o.contiguous())[:, :5], env, ctl, max_steps=300)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"N={N}: {dt:.2f} s for {N} episodes ({int(m['steps'].sum())} env steps, mean {m['steps'].mean():.0f} steps, "
              f"early-stop {m['stopped_early'].mean() * 100:.0f}%) -> {N / dt:.0f} episodes/s, {m['steps'].sum() / dt / 1e6:.2f} M env-steps/s")


if __name__ == "__main__":
    main()
