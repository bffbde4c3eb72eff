#!/usr/bin/env python3
"""Would the BPTTs of two stacked h = 256 layers overlap if they ran on two streams?  Two independent uav_lstm_bwd calls
(own handles / workspaces) on one stream vs on two streams; ms per pair of T-step passes."""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo._lib import lib, check  # noqa: E402


def main():
    dev, N, T, H = torch.device("cuda:0"), 4096, 64, 256
    h1 = ops.Context.get(dev).handle
    h2 = C.c_void_p()
    check(lib().uav_create(C.byref(h2), 0, 256 << 20), "uav_create")
    mk = lambda: dict(stash=torch.rand(N, T, 6 * H, device=dev) * 0.8 + 0.1, dy=torch.randn(N, T, H, device=dev) * 1e-6,
                      w=torch.randn(4 * H, H, device=dev) * 0.05, wi=torch.randn(4 * H, H, device=dev) * 0.05,
                      dg=torch.empty(N, T, 4 * H, device=dev), dx=torch.empty(N, T, H, device=dev))
    a, b = mk(), mk()

    def run(h, d, stream, with_dx):
        check(lib().uav_lstm_bwd(h, None, ops._p(d["stash"]), ops._p(d["w"]), ops._p(d["dy"]), None, None, 0, None, None, N, T, H,
                                 ops._p(d["dg"]), None, None, ops._p(d["wi"]) if with_dx else None, H,
                                 ops._p(d["dx"]) if with_dx else None, C.c_void_p(stream.cuda_stream)), "uav_lstm_bwd")

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("one stream", "two streams"):
        for rep in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s1.wait_event(e0); s2.wait_event(e0)
            run(h1, a, s1, True)
            run(h2, b, s1 if mode == "one stream" else s2, False)
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(s1); eb.record(s2)
            torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
            e1.record()
            torch.cuda.synchronize()
        print(f"{mode}: {e0.elapsed_time(e1):.2f} ms for two {T}-step backward passes ({e0.elapsed_time(e1) / T * 1e3:.1f} us per step pair)", flush=True)


if __name__ == "__main__":
    main()
