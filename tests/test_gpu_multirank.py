"""Two ranks on ONE GPU over gloo: the data-parallel trainer path end to end (env shards by global
index, advantage-stat all-reduce, flat-gradient all-reduce, replicated curriculum).  RCCL itself
needs one GPU per rank and is exercised by the driver's multi-GPU bench.  -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path[:0] = [ROOT, PKG]
from uavppo.trainer import VecPPOTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
N = 64 // world
kind = os.environ["POLICY"]
if kind == "lstm64":          # fused persistent rollout + sequence kernels
    tr = VecPPOTrainer(N, 32, "lstm", hidden=64, device="cuda:0", seed=11, rank=rank, world_size=world, epochs=2)
elif kind == "mlp":           # the reference's policy: step-wise rollout (uav_policy_sample keyed by global env index)
    tr = VecPPOTrainer(N, 32, "mlp", device="cuda:0", seed=11, rank=rank, world_size=world, epochs=2)
else:                         # C5 family: h=256 stacked x2 + trend obs, step-wise LSTM rollout
    tr = VecPPOTrainer(N, 12, "lstm", hidden=256, layers=2, trend_k=2, variant="v2.1", device="cuda:0", seed=11, rank=rank,
                       world_size=world, epochs=2)
tr.record = True
for _ in range(2):
    tr.train_iteration()
out = {"flat": tr.policy.flat.cpu(), "adv": tr.adv_n.cpu(), "obs": tr.buf["obs"].cpu(), "radius": tr.radius,
       "gn": [g.item() for _, g in tr.log], "hist": len(tr.curriculum.success_history)}
torch.save(out, os.environ["OUT"] + f".{rank}")
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _run(world, out, port, policy):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OUT=out,
                   POLICY=policy)
        code = f"ROOT={ROOT!r}; PKG={PKG!r}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0


@pytest.mark.parametrize("policy", ["lstm64", "mlp", "lstm256x2"])
def test_two_ranks_equal_one_rank(tmp_path, policy):
    """A job is invariant to its sharding on EVERY rollout path: the fused kernel and uav_policy_sample both key the
    action RNG by (seed; step, GLOBAL env index, iteration)."""
    port = 29600 + os.getpid() % 1000 + 7 * ["lstm64", "mlp", "lstm256x2"].index(policy)
    _run(1, str(tmp_path / "w1"), port, policy)
    _run(2, str(tmp_path / "w2"), port + 1, policy)
    one = torch.load(tmp_path / "w1.0")
    a, b = torch.load(tmp_path / "w2.0"), torch.load(tmp_path / "w2.1")
    # rollouts: shard r of the 2-rank job == envs [32r, 32r+32) of the 1-rank job (same global RNG keys)
    assert torch.equal(torch.cat([a["obs"], b["obs"]], 0), one["obs"])
    # whole-buffer advantage normalisation and the averaged gradient -> identical parameters on all ranks
    assert torch.allclose(torch.cat([a["adv"], b["adv"]], 0), one["adv"], atol=1e-5)
    assert torch.equal(a["flat"], b["flat"])
    # Adam turns a gradient g into a step of lr * g / (|g| + eps): an element whose gradient is nearly zero can move by a
    # different fraction of lr when the ranks' partial sums are added in another order.  Bound the difference by a tenth of
    # the largest possible movement (4 optimiser steps x lr) and require it to be negligible on average.
    diff = (a["flat"] - one["flat"]).abs()
    assert diff.max().item() < 0.1 * 4 * 3e-5 and diff.mean().item() < 2e-7
    assert np.allclose(a["gn"], one["gn"], rtol=1e-3) and a["gn"] == b["gn"]
    assert a["radius"] == b["radius"] == one["radius"] and a["hist"] == one["hist"]
