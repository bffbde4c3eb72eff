#!/usr/bin/env python3
"""Robustness sweep: random (N, T, H) shapes through rollout + one PPO gradient with the default (split-fp16) kernels and
with the exact-f32 kernels (uav_set_lstm_arith(UAV_ARITH_F32_MFMA)); the two gradients must agree to f32 noise and be finite.
(With a handful of samples the value gradient is (V_recomputed - V_rollout), pure rounding noise of whichever forward
produced it, so the relative test gets an absolute floor of 1e-7 / sqrt(samples).)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402


def main():
    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    worst = 0.0
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        H = (64, 128)[case % 2]
        N = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 100, 255, 256, 257, int(rng.randint(1, 600))]))
        T = int(rng.choice([1, 2, 7, 8, 9, 16, 31, 32, 33, 64, 100, int(rng.randint(1, 200))]))
        ops.set_lstm_arith("fp16x3")
        tr = VecPPOTrainer(N, T, "lstm", hidden=H, device="cuda:0", seed=case, use_curriculum=False)
        tr.radius = 80.0
        tr.collect()
        tr.compute_advantages()
        b, pol = tr.buf, tr.policy
        n = N * T
        args = (b["act"].reshape(-1), b["logp"].reshape(-1), tr.adv_n.reshape(-1), tr.ret.reshape(-1), b["val"].reshape(-1),
                1.0 / n, 0.2, 0.01)
        gs = []
        for f32 in (False, True):
            ops.set_lstm_arith("f32_mfma" if f32 else "fp16x3")
            heads = pol.heads(b["obs"], b["keep"], tr.h0, tr.c0, tr.work)
            loss = torch.zeros(4, dtype=torch.float64, device="cuda:0")
            dheads = torch.empty(n, 6, device="cuda:0")
            dbias = torch.empty(6, device="cuda:0")
            ops.ppo_loss_heads(heads, *args, loss, dheads, dbias)
            gs.append(pol.backward(dheads, tr.work, dbias).clone().double())
        ops.set_lstm_arith("fp16x3")
        # the rollout's own heads / stash against the recomputed forward
        hr = tr.work["heads"].reshape(n, 6) if tr._rollout_forward_valid else None
        dn = float((gs[0] - gs[1]).norm())
        rel = dn / (float(gs[1].norm()) + 1e-30)
        ok = np.isfinite(rel) and dn <= 5e-5 * float(gs[1].norm()) + 1e-7 / np.sqrt(n) and bool(torch.isfinite(gs[0]).all())
        if hr is not None:
            okh = bool(torch.allclose(hr, heads, atol=3e-5, rtol=1e-4))
            ok = ok and okh
        worst = max(worst, rel)
        print(f"case {case:3d} N={N:4d} T={T:4d} H={H:4d}  rel grad diff {rel:.2e}  {'ok' if ok else 'FAIL'}")
        if not ok:
            sys.exit(1)
    print("all ok; worst relative difference", worst)


if __name__ == "__main__":
    main()
