#!/usr/bin/env python3
"""BPTT of an h = 256 layer at BASELINE C5's per-GPU shape (4096 envs): persistent cluster kernel against the per-step launches
(cell_bwd_h3 + step_bwd_h3).  usage: perf_cluster_bwd.py [T=256] [N=4096]"""
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
lib = ops.lib()
for I, dx in ((8, False), (256, True)):
    k = 1 / 16.0
    keep = (torch.rand(N, T, generator=g) > 0.002).float().to(dev)
    stash = torch.rand(N, T, 6 * H, device=dev) * 0.8 + 0.1
    w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(dev)
    w_ih = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(dev)
    dy = torch.randn(N, T, H, device=dev) / (N * T)
    dgates = torch.empty(N, T, 4 * H, device=dev)
    dxo = torch.empty(N, T, H, device=dev) if dx else None
    dh0 = torch.empty(N, H, device=dev); dc0 = torch.empty(N, H, device=dev)

    def run():
        ops.check(lib.uav_lstm_bwd(ops._h(dy), ops._p(keep), ops._p(stash), ops._p(w_hh), ops._p(dy), None, None, 0, None, None, N, T, H,
                                   ops._p(dgates), ops._p(dh0), ops._p(dc0), ops._p(w_ih) if dx else None, H if dx else 8,
                                   ops._p(dxo) if dx else None, ops._stream()), "uav_lstm_bwd")
    for mode, fl in (("per-step", []), ("cluster", ["cluster"]), ("cluster-sc1", ["cluster", "cluster_sc1"]), ("c-nowait", ["cluster", "abl_wait"]), ("c-nostores", ["cluster", "abl_stash"]),
                     ("c-nofetch", ["cluster", "abl_wait", "abl_fetch"]), ("c-nomfma", ["cluster", "abl_mfma"])):
        ops.set_debug_flags(*fl)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{'with dx' if dx else 'no dx  '} {mode:11s}: {ms:8.3f} ms per pass = {1e3 * ms / T:6.2f} us per step   (N={N}, T={T})", flush=True)
    ops.set_debug_flags()
    if os.environ.get("UAVPPO_LIB", "").endswith("libuavppo_prof.so"):        # instrumented build: cycles per phase, wave 0 of workgroup 0
        import ctypes
        ops.set_debug_flags("cluster")
        run()
        torch.cuda.synchronize()
        ops.set_debug_flags()
        out = (ctypes.c_ulonglong * 12)()
        fn = lib.uav_c8_profile
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        fn(ops.Context.get(torch.device(dev)).handle, out)
        names = ["cell backward (incl. wait for the stash rows)", "dgates stores", "scale + pieces -> LDS", "barrier 1", "products + partial stores", "publish vmcnt(0)",
                 "barrier 2", "flag + next rows issue + poll", "barrier 3", "gather + sums (+ dx store)"]
        steps = T * (((N + 63) // 64 + 31) // 32)
        print(f"{'with dx' if dx else 'no dx'}: cycles per tile-step of wave 0, workgroup 0 ({steps} tile-steps; total {sum(out) / steps:.0f}):")
        for nm, v in zip(names, out):
            print(f"   {nm:46s} {v / steps:8.0f}")
print("cluster wait time-outs:", ops.lstm_cluster_errors())
