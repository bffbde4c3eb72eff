#!/usr/bin/env python3
"""Same process, same buffers: the C3 sequence kernels with the weight-gradient kernel's slabs interleaved over the blocks (default)
and in contiguous runs per block (UAV_DEBUG_WGRAD_CONTIG), alternating.  -> avg ms of fwd / bwd / wgrad / rollout and the iteration.

NOT runnable against the tree: the experiment's patch (measured, no gain, not kept -- profiles/r04_z_wgrad_mapping_ab.log) gave
lstm_wgrad_h3_kernel an `interleave` argument (r_begin = blockIdx.x * KS6, slab step = gridDim.x * KS6 rows instead of
r_begin = blockIdx.x * rows_per_block, step KS6) behind a debug bit 0x10 "wgrad_contig" in uav_set_debug_flags / ops.DEBUG_FLAGS."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
import torch  # noqa: E402
from uavppo import ops  # noqa: E402
from uavppo.trainer import VecPPOTrainer  # noqa: E402

tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", use_curriculum=False, seed=3)
for _ in range(3):
    tr.train_iteration()
for rnd in range(3):
    for name, flags in (("interleaved", ()), ("contiguous", ("wgrad_contig",))):
        ops.set_debug_flags(*flags)
        ops.KERNEL_TIMER.enable(("lstm_fwd", "lstm_bwd", "lstm_wgrad", "rollout"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            tr.train_iteration()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 12
        s = ops.KERNEL_TIMER.summary()
        print(f"{name:12s}", " ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in sorted(s.items())), "iteration %.3f ms" % (dt * 1e3))
ops.set_debug_flags()
