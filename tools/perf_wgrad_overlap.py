#!/usr/bin/env python3
"""Would the weight-gradient passes of two stacked h = 256 layers overlap on two streams (own handles / workspaces)?"""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops  # noqa: E402
from uavppo._lib import lib, check  # noqa: E402


def main():
    dev, N, T, H = torch.device("cuda:0"), 4096, 256, 256
    h1 = ops.Context.get(dev).handle
    h2 = C.c_void_p()
    check(lib().uav_create(C.byref(h2), 0, 256 << 20), "uav_create")
    def mk(I):
        return dict(I=I, x=torch.randn(N, T, I, device=dev) * 0.5, h0=torch.zeros(N, H, device=dev), y=torch.rand(N, T, H, device=dev),
                    st=torch.rand(N, T, 6 * H, device=dev), dg=torch.randn(N, T, 4 * H, device=dev) * 1e-6,
                    wi=torch.randn(4 * H, I, device=dev) * 0.05, dwi=torch.empty(4 * H, I, device=dev), dwh=torch.empty(4 * H, H, device=dev),
                    db=torch.empty(4 * H, device=dev))
    a, b = mk(256), mk(8)

    def run(h, d, stream):
        check(lib().uav_lstm_wgrad(h, ops._p(d["x"]), None, ops._p(d["h0"]), ops._p(d["y"]), ops._p(d["st"]), ops._p(d["dg"]), ops._p(d["wi"]),
                                   None, 0, N, T, d["I"], H, ops._p(d["dwi"]), ops._p(d["dwh"]), ops._p(d["db"]), None, None, None,
                                   C.c_void_p(stream.cuda_stream)), "uav_lstm_wgrad")

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("one stream", "two streams"):
        for rep in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s1.wait_event(e0); s2.wait_event(e0)
            run(h1, a, s1)
            run(h2, b, s1 if mode == "one stream" else s2)
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(s1); eb.record(s2)
            torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
            e1.record()
            torch.cuda.synchronize()
        print(f"{mode}: {e0.elapsed_time(e1):.2f} ms for the two layers' weight-gradient passes", flush=True)


if __name__ == "__main__":
    main()
