#!/usr/bin/env python3
"""One layer of the h = 256 step path: uav_lstm_fwd over T steps at N = 4096 (I = 8 and I = 256), us per step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402

def main():
    dev, N, T, H = "cuda:0", int(os.environ.get("STEP_N", 4096)), 64, 256
    for I in (8, 256):
        x = torch.randn(N, T, I, device=dev) * 0.5
        h0, c0 = torch.zeros(N, H, device=dev), torch.zeros(N, H, device=dev)
        w_ih, w_hh = torch.randn(4 * H, I, device=dev) * 0.05, torch.randn(4 * H, H, device=dev) * 0.05
        b = torch.zeros(4 * H, device=dev)
        stash, y = torch.empty(N, T, 6 * H, device=dev), torch.empty(N, T, H, device=dev)
        for _ in range(2):
            ops.lstm_fwd(x, None, h0, c0, w_ih, w_hh, b, b, stash=stash, y=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.lstm_fwd(x, None, h0, c0, w_ih, w_hh, b, b, stash=stash, y=y)
        e1.record(); torch.cuda.synchronize()
        print(f"I={I}: {e0.elapsed_time(e1) / 3 / T * 1e3:.1f} us per step (launch gaps included)", flush=True)

if __name__ == "__main__":
    main()
