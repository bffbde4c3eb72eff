"""Per-step time of the h = 256 BPTT step launches (uav_lstm_bwd alone, one layer, one stream) at C5's per-GPU shape.
UAVPPO_LIB selects an instrumented / ablation build (tools/experiments/ab_bptt.sh, on the experiment patch).  Usage: python tools/perf_bptt.py [N] [T]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd"))
from uavppo import ops
from uavppo._lib import lib
from uavppo.ops import _h, _p, _stream, check, F32

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = 256
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x = r(N, T, H) * 0.5
w_ih, w_hh = r(4 * H, H) * 0.05, r(4 * H, H) * 0.05
b = r(4 * H) * 0.05
h0 = torch.zeros(N, H, device=dev)
out = ops.lstm_fwd(x, None, h0, h0, w_ih, w_hh, b, b, want_stash=True)
stash = out["stash"] if isinstance(out, dict) else out[-1]
dheads = r(N, T, 6) * 1e-3
w_head = r(6, H) * 0.1
dy = r(N, T, H) * 1e-3
dgates = ops.lstm_dgates(N, T, H, dev)
dx = torch.empty(N, T, H, device=dev)


def run(kind):
    a = dict(dy=None, dheads=None, w_head=None, nh=0, w_ih=None, dx=None)
    if kind == "dheads":
        a.update(dheads=dheads, w_head=w_head, nh=6)
    elif kind == "dy":
        a.update(dy=dy)
    else:
        a.update(dy=dy, w_ih=w_ih, dx=dx)
    check(lib().uav_lstm_bwd(_h(x), None, _p(stash), _p(w_hh), _p(a["dy"]), _p(a["dheads"]), _p(a["w_head"]), a["nh"], None, None,
                             N, T, H, _p(dgates), None, None, _p(a["w_ih"]), H, _p(a["dx"]), _stream()), "uav_lstm_bwd")


for kind in ("dheads", "dy", "dy+dx"):
    run(kind)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run(kind)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('UAVPPO_LIB', 'libuavppo.so').split('/')[-1]:28s} {kind:8s} N={N} T={T}: {e0.elapsed_time(e1) / 3 / T * 1e3:7.2f} us per step")
