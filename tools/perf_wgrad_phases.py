#!/usr/bin/env python3
"""Phase cycles of lstm_wgrad_h3_kernel<128> at C3's shape (UAVPPO_LIB=tools/libuavppo_prof.so, built by tools/build_prof.sh):
per slab, for an EARLY wave (w = 0) and a LATE wave (w = 7) of workgroup 0.  Phases (s_memtime deltas summed over the slabs):
0 loop top -> 2 mfma pair 0 -> 5 commit_b + load_b (next B planes to LDS, B loads of slab + 2) -> 3 split pair 1 (+ A loads of
slab + 1, pair 1) -> 4 mfma pair 1 -> 7 slab barrier -> 1 split pair 0 of the next slab (+ its A loads)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo import ops  # noqa: E402
from uavppo._lib import lib  # noqa: E402

dev = "cuda:0"
N, T, I, H, NH = 4096, 128, 6, 128, 6
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, T, I, device=dev, generator=g)
keep = (torch.rand(N, T, device=dev, generator=g) > 0.02).float()
h0 = torch.zeros(N, H, device=dev)
y = torch.rand(N, T, H, device=dev, generator=g) * 2 - 1
stash = torch.rand(N, T, 6 * H, device=dev, generator=g)
dg = torch.randn(N, T, 4 * H, device=dev, generator=g) * 1e-4
dheads = torch.randn(N, T, NH, device=dev, generator=g) * 1e-3
w_ih = torch.randn(4 * H, I, device=dev, generator=g) * 0.1
for _ in range(3):
    ops.lstm_wgrad(x, keep, h0, y, stash, dg, w_ih, dheads=dheads)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.lstm_wgrad(x, keep, h0, y, stash, dg, w_ih, dheads=dheads)
e1.record()
torch.cuda.synchronize()
print(f"uav_lstm_wgrad (kernel + reduce): {e0.elapsed_time(e1) / 10:.3f} ms")
out = (C.c_ulonglong * 16)()
f = lib().uav_wx6_prof_read
f.argtypes, f.restype = [C.c_void_p], C.c_int
assert f(out) == 0
nslab = N * T // 256 // 32
names = {6: "(experiment) wait for all loads", 0: "loop top", 2: "mfma pair 0", 5: "commit_b + load_b", 3: "split pair 1 + A loads", 4: "mfma pair 1", 7: "slab barrier", 1: "split pair 0 (next) + A loads"}
for wv, label in ((0, "early wave 0"), (1, "late wave 7")):
    v = [out[wv * 8 + i] for i in range(8)]
    print(f"{label}: {sum(v) / nslab:8.0f} counter ticks per slab ({nslab} slabs)")
    for i in (0, 2, 6, 5, 3, 4, 7, 1):
        print(f"    {names[i]:34s} {v[i] / nslab:8.0f}")
