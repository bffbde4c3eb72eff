#!/usr/bin/env python3
"""Phase cycles of mlp_ppo_grad_kernel (instrumented build: bash tools/build_prof.sh; UAVPPO_LIB=tools/libuavppo_prof.so)
and wall time of the fused MLP kernels at the C3 buffer shape."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo import _lib, ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

N, T = 4096, 128
tr = VecPPOTrainer(N, T, "mlp", device="cuda:0", seed=1, use_curriculum=False)
tr.collect()
tr.compute_advantages()
b = tr.buf
n = N * T
args = (tr.policy.flat, b["obs"].reshape(n, 6), b["act"].reshape(-1), b["logp"].reshape(-1), tr.adv_n.reshape(-1),
        tr.ret.reshape(-1), b["val"].reshape(-1), 1.0 / n, 0.2, 0.01, tr.loss_sums, tr.policy.grad)
for _ in range(3):
    ops.mlp_ppo_grad(*args)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    ops.mlp_ppo_grad(*args)
ev[1].record()
torch.cuda.synchronize()
print("mlp_ppo_grad: %.3f ms per call (%d samples)" % (ev[0].elapsed_time(ev[1]) / 10, n))
ev[0].record()
for _ in range(5):
    tr.collect()
ev[1].record()
torch.cuda.synchronize()
print("rollout_mlp: %.3f ms per rollout" % (ev[0].elapsed_time(ev[1]) / 5))
lib = _lib.lib()
if hasattr(lib, "uav_mlp_prof_read"):
    ops.mlp_ppo_grad(*args)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    lib.uav_mlp_prof_read(out)
    names = ["stage X", "L1+LN1", "L2+LN2", "heads fwd", "loss", "heads bwd+LN2 bwd", "dW2", "da1", "LN1 bwd", "dW1"]
    ntile = (n // 32 + 255) // 256
    for wv in range(2):
        tot = sum(out[wv * 16 + i] for i in range(10))
        print("wave %d: %d cycles per tile" % (wv, tot // ntile))
        for i, nm in enumerate(names):
            print("   %-20s %7d" % (nm, out[wv * 16 + i] // ntile))
