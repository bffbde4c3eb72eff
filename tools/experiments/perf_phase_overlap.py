"""Do a product-only BPTT launch chain and an epilogue-only one overlap when they share the GPU?  Two ablation builds of
libuavppo (tools/ab_bptt.sh: a7 = products + residual epilogue without its HBM traffic, a8 = epilogue without products), each
with its own handle, run alone and then together on two streams."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import _lib
dev, N, T, H = torch.device("cuda:0"), 4096, 64, 256
torch.zeros(1, device=dev)
libs = {}
for tag in ("a7", "a8"):
    L = C.CDLL(os.path.join(ROOT, "tools", "bin", f"libuavppo_bptt_{tag}.so"))
    for name, (res, args) in _lib.SIGNATURES.items() if hasattr(_lib, "SIGNATURES") else []:
        pass
    L.uav_create.restype = C.c_int; L.uav_lstm_bwd.restype = C.c_int
    h = C.c_void_p()
    assert L.uav_create(C.byref(h), 0, C.c_size_t(256 << 20)) == 0
    libs[tag] = (L, h)
P = lambda t: C.c_void_p(t.data_ptr())
mk = lambda: dict(stash=torch.rand(N, T, 6 * H, device=dev) * 0.8 + 0.1, dy=torch.randn(N, T, H, device=dev) * 1e-6,
                  w=torch.randn(4 * H, H, device=dev) * 0.05, dg=torch.empty(N, T, 4 * H, device=dev))
d = {"a7": mk(), "a8": mk()}
def run(tag, stream):
    L, h = libs[tag]; x = d[tag]
    rc = L.uav_lstm_bwd(h, None, P(x["stash"]), P(x["w"]), P(x["dy"]), None, None, 0, None, None, N, T, H, P(x["dg"]), None, None, None, H, None,
                        C.c_void_p(stream.cuda_stream))
    assert rc == 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("a7 alone", "a8 alone", "both, two streams"):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s1.wait_event(e0); s2.wait_event(e0)
        if mode != "a8 alone": run("a7", s1)
        if mode != "a7 alone": run("a8", s2)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(s1); eb.record(s2)
        torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
        e1.record(); torch.cuda.synchronize()
    print(f"{mode}: {e0.elapsed_time(e1) / T * 1e3:.1f} us per step", flush=True)
