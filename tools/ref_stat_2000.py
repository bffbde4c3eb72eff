#!/usr/bin/env python3
"""The reference's own training statistic on the reference's run length:
PPOV2.0/training_results2_0.csv = 2000 episodes, 1299 successes (65 %), 1,142,500 env steps, final curriculum radius 8.386
(V2.1: 1272 / 1,145,498 / 8.276) -- one unseeded run of train_ppo() (MLP policy, lr 3e-5, 256-step buffer, 5 epochs).

This runs the drop-in script's `train_ppo_vectorised(episodes=2000)` with the reference's hyper-parameters and tallies the
CSV rows the reference's way (sum of Success, sum of Steps, Current_Radius of the last row), for several seeds and buffer
shapes with NUM_ENVS x HORIZON = 256 samples per update, i.e. the reference's number of optimiser steps per env step:
   1 x 256  the reference's exact buffer semantics (one env, a flat 256-step stream across episode ends)
   8 x 32   eight envs, 32-step rollouts
The curriculum differs from the reference in one documented way: radius / bonus take effect at the next ROLLOUT (<= HORIZON
steps late) instead of at the next episode.    usage: ref_stat_2000.py [variant v2.0|v2.1] [seeds] [shapes e.g. 1x256,8x32]"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")
sys.path[:0] = [ROOT, PKG]
import config  # noqa: E402

REF = {"v2.0": (1299, 1142500, 8.386), "v2.1": (1272, 1145498, 8.276), "v1.1": (1283, 1098470, float("nan"))}


def run(variant, seed, n, t):
    for k, v in dict(NUM_ENVS=n, HORIZON=t, POLICY="mlp", ENV_VARIANT=variant, SEED=seed, LEARNING_RATE=3e-5, EPOCHS=5,
                     NUM_MINIBATCHES=1).items():
        setattr(config, k, v)
    spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(PKG, "train_ppo2.0.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    t0 = time.perf_counter()
    tr, rows = m.train_ppo_vectorised(episodes=2000, csv_path=None, model_path=None, log_every=0)
    dt = time.perf_counter() - t0
    succ = sum(int(r[2]) for r in rows)
    steps = sum(int(r[8]) for r in rows)
    return succ, steps, float(rows[-1][10]), tr.iteration, dt


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "v2.0"
    seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4").split(",")]
    shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[3] if len(sys.argv) > 3 else "1x256,8x32").split(",")]
    r = REF[variant]
    print(f"reference ({variant}, one unseeded run): successes {r[0]} / 2000 ({100 * r[0] / 2000:.0f} %), total steps {r[1]:,}, final radius {r[2]}")
    for n, t in shapes:
        for seed in seeds:
            succ, steps, radius, iters, dt = run(variant, seed, n, t)
            print(f"build  {variant} {n} env x {t} steps, seed {seed}: successes {succ} / 2000 ({100 * succ / 2000:.0f} %), total steps {steps:,}, "
                  f"final radius {radius:.3f}   [{iters} updates of 256 samples x 5 epochs, {dt:.1f} s]", flush=True)


if __name__ == "__main__":
    main()
