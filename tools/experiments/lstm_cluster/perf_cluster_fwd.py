#!/usr/bin/env python3
"""uav_lstm_fwd at h = 256 (BASELINE C5's per-GPU shape: 4096 envs): persistent cluster kernel against the per-step launches.
usage: perf_cluster_fwd.py [T=256] [N=4096]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H, dev = 256, "cuda:0"
g = torch.Generator("cpu").manual_seed(0)
for I in (8, 256):
    k = 1 / 16.0
    x = torch.randn(N, T, I, generator=g).to(dev)
    keep = (torch.rand(N, T, generator=g) > 0.002).float().to(dev)
    h0 = torch.zeros(N, H, device=dev); c0 = torch.zeros(N, H, device=dev)
    w_ih = ((torch.rand(4 * H, I, generator=g) * 2 - 1) * k).to(dev); w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(dev)
    b = ((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(dev)
    y = torch.empty(N, T, H, device=dev); stash = torch.empty(N, T, 6 * H, device=dev)
    MODES = {"per-step": None, "cluster": [], "c-nowait": ["abl_wait"], "c-nostash": ["abl_stash"], "c-nofetch": ["abl_wait", "abl_fetch"],
             "c-nomfma": ["abl_mfma"], "c-nowait-nostash": ["abl_wait", "abl_stash"], "c-only-mfma": ["abl_wait", "abl_stash", "abl_fetch"],
             "c-nothing": ["abl_wait", "abl_stash", "abl_fetch", "abl_mfma"]}
    if len(sys.argv) > 3 and sys.argv[3] == "short":
        MODES = {"per-step": None, "cluster": [], "cluster-sc1": ["cluster_sc1"]}
    for mode, fl in MODES.items():
        ops.set_debug_flags(*([] if fl is None else ["cluster"] + fl))
        for _ in range(2):
            ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b, b, stash=stash, y=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 3
        for _ in range(reps):
            ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b, b, stash=stash, y=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"I={I:3d} {mode:17s}: {ms:8.3f} ms per pass = {1e3 * ms / T:6.2f} us per step   (N={N}, T={T})", flush=True)
    ops.set_debug_flags()
    if os.environ.get("UAVPPO_LIB", "").endswith("libuavppo_prof.so"):        # instrumented build: cycles per phase, wave 0 of workgroup 0
        import ctypes
        from uavppo import _lib
        ops.set_debug_flags("cluster")
        ops.lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b, b, stash=stash, y=y)
        torch.cuda.synchronize()
        ops.set_debug_flags()
        out = (ctypes.c_ulonglong * 12)()
        fn = _lib.lib().uav_c8_profile
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        fn(ops.Context.get(torch.device(dev)).handle, out)
        names = ["top: masks + x loads", "products + cell + h stores", "peer fetch: issue", "peer fetch: latency", "publish vmcnt(0)", "barrier 2", "flag + x planes",
                 "poll", "barrier 3", "peer fetch: LDS stores", "stash stores", "barrier 4"]
        steps = T * (((N + 63) // 64 + 31) // 32)
        tot = sum(out)
        print(f"I={I}: cycles per tile-step of wave 0, workgroup 0 ({steps} tile-steps; total {tot / steps:.0f}):")
        for nm, v in zip(names, out):
            print(f"   {nm:36s} {v / steps:8.0f}")
print("cluster wait time-outs:", ops.lstm_cluster_errors())
