#!/usr/bin/env python3
"""Per-kernel times of one PPO epoch at the C3 shape (HIP events around each ABI call, bursts of epochs so the
clocks settle) and the gradient difference against the exact-f32 kernels when run with UAV_LSTM_F32_MFMA=1."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402


def main():
    N, T, H = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (4096, 128, 128)))
    dev = "cuda:0"
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, device=dev, use_curriculum=False, seed=3)
    tr.collect()
    tr.compute_advantages()
    pol, b = tr.policy, tr.buf
    heads = pol.heads(b["obs"], b["keep"], tr.h0, tr.c0, tr.work)
    g = torch.Generator(device=dev).manual_seed(1)
    dheads = torch.randn(heads.shape, generator=g, device=dev) / heads.shape[0]
    ops.KERNEL_TIMER.enable(("lstm_fwd", "lstm_bwd", "lstm_wgrad"))
    for _ in range(30):
        pol.heads(b["obs"], b["keep"], tr.h0, tr.c0, tr.work)
        pol.backward(dheads, tr.work, tr.dhead_bias)
    torch.cuda.synchronize()
    for k, s in ops.KERNEL_TIMER.summary().items():
        print("%-12s: %.3f ms avg over %d" % (k, s["avg_ms"], s["n"]))
    from uavppo import _lib
    L = _lib.lib()
    if hasattr(L, "uav_wx6_prof_read"):             # instrumented build (tools/build_prof.sh)
        import ctypes as C
        buf = (C.c_ulonglong * 16)()
        L.uav_wx6_prof_read(buf)
        ns = N * T // 256 // 32
        names = ["loop+load_b", "split0+load_a", "mfma0", "split1+load_a", "mfma1", "heads", "commit", "barrier"]
        for wv in range(2):
            print("  wgrad wave %s cycles/slab: " % ("0" if wv == 0 else "last") +
                  " | ".join("%s %.0f" % (names[i], buf[wv * 8 + i] / ns) for i in range(8)) +
                  " | total %.0f" % (sum(buf[wv * 8 + i] for i in range(8)) / ns))
    if hasattr(L, "uav_x6_prof_read"):
        buf = (C.c_ulonglong * 16)()
        L.uav_x6_prof_read(buf)
        names = ["loop", "wait+ring reads+dma issue", "barrier b2", "dy+pointwise+split+stores", "mfma+partials", "barrier b1", "sum"]
        for wv in range(2):
            print("  bwd wave %s cycles/step: " % ("0" if wv == 0 else "last") +
                  " | ".join("%s %.0f" % (names[i], buf[wv * 8 + i] / T) for i in range(7)) +
                  " | total %.0f" % (sum(buf[wv * 8 + i] for i in range(7)) / T))
    grad = pol.grad.detach().cpu().double().numpy()
    out = os.path.join(ROOT, "gpurun_out", "grad_%s.npy" % ("f32" if os.environ.get("UAV_LSTM_F32_MFMA") else "x6"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.save(out, grad)
    other = out.replace("_x6", "_F").replace("_f32", "_x6").replace("_F", "_f32")
    if os.path.exists(other):
        o = np.load(other)
        print("grad diff vs %s: max abs %.3e, rel-to-norm %.3e" % (os.path.basename(other), np.abs(grad - o).max(),
                                                                   np.linalg.norm(grad - o) / np.linalg.norm(o)))


if __name__ == "__main__":
    main()
