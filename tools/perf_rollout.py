"""Rollout kernel timing; with the instrumented build (tools/build_prof.sh, UAVPPO_LIB=tools/libuavppo_prof.so) also the
phase split of wave 0 per step."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uav-wrf-les-ppo-lstm_amd"))
from uavppo import _lib, ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

N, T, H = 4096, 128, 128
tr = VecPPOTrainer(N, T, "lstm", hidden=H, device="cuda:0", seed=3, use_curriculum=False)
for _ in range(3):
    tr.collect()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    tr.collect()
e1.record()
torch.cuda.synchronize()
print(f"collect: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
L = _lib.lib()
if hasattr(L, "uav_roll_prof_read"):
    buf = (ctypes.c_ulonglong * 8)()
    L.uav_roll_prof_read(buf)
    names = ["finish_cell", "barrier1 wait", "heads", "softmax+sample", "env step", "park+reset+obs", "rng+wind (shadow)", "barrier2+acc0+fixup"]
    tot = sum(buf)
    for n, v in zip(names, buf):
        print(f"  {n:18s} {v / (T + 1):8.0f} cycles/step  {100.0 * v / tot:5.1f} %")
    print(f"  total {tot / (T + 1):.0f} cycles/step (s_memtime ticks at 100 MHz if constant-rate: compare with the wall time)")
