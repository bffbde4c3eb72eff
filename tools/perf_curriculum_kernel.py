"""Duration of uav_curriculum_update on an IDLE GPU (HIP events), for messages of 6000 and 16384 ended episodes from 1 and 8
ranks.  Inside a training iteration rocprofv3 shows the same launch at ~300 us: there it runs on the side stream next to
full-chip kernels of the update, and the one-workgroup dispatch waits for a free slot."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "uav-wrf-les-ppo-lstm_amd"))
from uavppo import ops                                                      # noqa: E402
from uavppo.dist_utils import SUCC_CAP                                      # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(3)
for world, cnt in ((1, 6000), (1, SUCC_CAP), (8, 6000), (8, SUCC_CAP)):
    m = np.zeros((world, 4 + SUCC_CAP + 1), np.uint8)
    for r in range(world):
        m[r, :4] = np.frombuffer(np.int32(cnt).tobytes(), np.uint8)
        m[r, 4:4 + cnt] = rng.random(cnt) < 0.5
    msgs = torch.from_numpy(m).to(dev)
    st = ops.curriculum_state(dev)
    for _ in range(3):
        ops.curriculum_update(st, msgs, SUCC_CAP)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.curriculum_update(st, msgs, SUCC_CAP)
    e1.record()
    torch.cuda.synchronize()
    print(f"world {world}  {cnt:6d} episodes per rank: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per launch (idle GPU, back to back)")
