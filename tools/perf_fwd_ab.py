#!/usr/bin/env python3
"""Times uav_lstm_fwd / bwd / wgrad at the C3 shape and prints the max difference against a torch f64 LSTM
(run with and without UAV_LSTM_F32_MFMA=1 / UAV_LSTM_BF16X6=1 to A/B the split-fp16 kernels against the exact-f32 and
split-bf16 ones)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402


def timeit(fn, n=8, burst=25):
    """median / min per-launch time over n bursts of `burst` back-to-back launches (clocks settle in a burst)."""
    for _ in range(burst):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(burst):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / burst)
    return float(np.median(ts)), float(np.min(ts))


def main():
    dev = "cuda:0"
    for (N, T, H) in ((4096, 128, 128), (256, 64, 64)):
        g = torch.Generator(device=dev).manual_seed(0)
        I = 6
        k = 1.0 / H ** 0.5
        w_ih = (torch.rand(4 * H, I, generator=g, device=dev) * 2 - 1) * k
        w_hh = (torch.rand(4 * H, H, generator=g, device=dev) * 2 - 1) * k
        b_ih = (torch.rand(4 * H, generator=g, device=dev) * 2 - 1) * k
        b_hh = (torch.rand(4 * H, generator=g, device=dev) * 2 - 1) * k
        x = torch.randn(N, T, I, generator=g, device=dev)
        keep = (torch.rand(N, T, generator=g, device=dev) > 0.02).float()
        h0 = torch.randn(N, H, generator=g, device=dev) * 0.1
        c0 = torch.randn(N, H, generator=g, device=dev) * 0.1
        y, hn, cn, stash = ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh)
        # f64 reference on a slice of envs
        M = 64
        hd, cd = h0[:M].double(), c0[:M].double()
        W, U, bb = w_ih.double(), w_hh.double(), (b_ih + b_hh).double()
        err = 0.0
        for t in range(T):
            kk = keep[:M, t:t + 1].double()
            hd, cd = hd * kk, cd * kk
            gts = x[:M, t].double() @ W.T + hd @ U.T + bb
            i_, f_, g_, o_ = gts.chunk(4, 1)
            cd = torch.sigmoid(f_) * cd + torch.sigmoid(i_) * torch.tanh(g_)
            hd = torch.sigmoid(o_) * torch.tanh(cd)
            err = max(err, float((y[:M, t].double() - hd).abs().max()))
        mode = "f32-mfma" if os.environ.get("UAV_LSTM_F32_MFMA") else ("bf16x6" if os.environ.get("UAV_LSTM_BF16X6") else "fp16x3")
        print(f"[{mode}] N={N} T={T} H={H}: max |y - f64| = {err:.3e}")
        print("   fwd (stash)  : %.3f ms (min %.3f)" % timeit(lambda: ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, stash=stash, y=y)))
        from uavppo import _lib
        L = _lib.lib()
        if hasattr(L, "uav_x6_prof_read"):          # instrumented build (UAVPPO_LIB=tools/libuavppo_prof.so)
            import ctypes as C
            buf = (C.c_ulonglong * 8)()
            L.uav_x6_prof_read(buf)
            for wv in range(2):
                c = [buf[wv * 4 + i] / T for i in range(4)]
                print("   wave %s cycles/step (s_memtime, 100 MHz ticks?): loads+mfma %.0f | gates+stores %.0f | barrier %.0f | loop %.0f | total %.0f"
                      % ("0" if wv == 0 else "last", c[1], c[2], c[3], c[0], sum(c)))


if __name__ == "__main__":
    main()
