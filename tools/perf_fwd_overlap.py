#!/usr/bin/env python3
"""Would the forward passes of two stacked h = 256 layers overlap on two streams?  Two independent uav_lstm_fwd calls (I = 8 and
I = 256, own handles / workspaces) on one stream vs on two streams."""
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
    def mk(I):
        return dict(I=I, x=torch.randn(N, T, I, device=dev) * 0.5, h0=torch.zeros(N, H, device=dev), c0=torch.zeros(N, H, device=dev),
                    wi=torch.randn(4 * H, I, device=dev) * 0.05, wh=torch.randn(4 * H, H, device=dev) * 0.05, b=torch.zeros(4 * H, device=dev),
                    y=torch.empty(N, T, H, device=dev), st=torch.empty(N, T, 6 * H, device=dev), hn=torch.empty(N, H, device=dev),
                    cn=torch.empty(N, H, device=dev))
    a, b = mk(8), mk(256)

    def run(h, d, stream):
        check(lib().uav_lstm_fwd(h, ops._p(d["x"]), None, ops._p(d["h0"]), ops._p(d["c0"]), ops._p(d["wi"]), ops._p(d["wh"]), ops._p(d["b"]),
                                 ops._p(d["b"]), N, T, d["I"], H, ops._p(d["y"]), ops._p(d["hn"]), ops._p(d["cn"]), ops._p(d["st"]), None, None, 0,
                                 None, C.c_void_p(stream.cuda_stream)), "uav_lstm_fwd")

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
        print(f"{mode}: {e0.elapsed_time(e1):.2f} ms for two {T}-step forward passes ({e0.elapsed_time(e1) / T * 1e3:.1f} us per step pair)", flush=True)


if __name__ == "__main__":
    main()
