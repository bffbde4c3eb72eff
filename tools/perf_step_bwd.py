#!/usr/bin/env python3
"""One layer of the h = 256 step path backwards: uav_lstm_bwd over T steps at N = 4096, us per step (cell + step kernels)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo._lib import lib, check  # noqa: E402

def main():
    dev, N, T, H = "cuda:0", int(os.environ.get("STEP_N", 4096)), 64, 256
    stash = torch.rand(N, T, 6 * H, device=dev) * 0.8 + 0.1
    dy = torch.randn(N, T, H, device=dev) * 1e-6
    w_hh = torch.randn(4 * H, H, device=dev) * 0.05
    dg = ops.lstm_dgates(N, T, H, dev)
    def run():
        check(lib().uav_lstm_bwd(ops._h(dy), None, ops._p(stash), ops._p(w_hh), ops._p(dy), None, None, 0, None, None, N, T, H,
                                 ops._p(dg), None, None, None, 0, None, ops._stream()), "uav_lstm_bwd")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"bwd: {e0.elapsed_time(e1) / 3 / T * 1e3:.1f} us per step (cell + step kernel, launch gaps included)", flush=True)

if __name__ == "__main__":
    main()
