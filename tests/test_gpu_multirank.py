"""Two ranks on ONE GPU over gloo: the data-parallel trainer path end to end (env shards by global
index, advantage-stat all-reduce, flat-gradient all-reduce, replicated curriculum).  RCCL itself
needs one GPU per rank: the driver's multi-GPU bench runs it across ranks, and test_rccl_single_rank_path runs every
exchange of the trainer through a ONE-rank RCCL communicator on this box's GPU.  -m gpu."""
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
backend = os.environ.get("BACKEND", "gloo")
if world > 1 or backend == "nccl":
    dist.init_process_group(backend, rank=rank, world_size=world)
if os.environ.get("COLLECTIVES") == "abi":      # the three exchanges on the C ABI's own RCCL communicator (uav_allreduce & co)
    from uavppo import dist_utils, ops
    dist_utils.use_abi_collectives(rank, world, "cuda:0")
    assert ops.comm_world("cuda:0") == world and ops.rccl_version() > 20000
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
       "gn": [g.item() for _, g in tr.log], "hist": len(tr.curriculum.success_history), "losses": [float(v) for v in tr.losses()]}
torch.save(out, os.environ["OUT"] + f".{rank}")
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
'''


def _run(world, out, port, policy, **extra):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OUT=out,
                   POLICY=policy, **extra)
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


@pytest.mark.parametrize("policy", ["lstm64", "mlp"])
def test_rccl_single_rank_path(tmp_path, policy):
    """backend "nccl" (= RCCL) with one rank and UAVPPO_FORCE_COLLECTIVES=1: the advantage-statistics and gradient all-reduces,
    the success-bit all-gather on the side stream and the loss / NaN-count all-reduce all go through a real RCCL
    communicator on the GPU, in the trainer's own stream choreography; a sum over one rank changes nothing, so the run
    must be BIT-identical to the same job without a process group."""
    port = 29700 + os.getpid() % 1000
    _run(1, str(tmp_path / "plain"), port, policy)
    _run(1, str(tmp_path / "rccl"), port + 1, policy, BACKEND="nccl", UAVPPO_FORCE_COLLECTIVES="1")
    a, b = torch.load(tmp_path / "plain.0"), torch.load(tmp_path / "rccl.0")
    for k in ("flat", "adv", "obs"):
        assert torch.equal(a[k], b[k]), k
    assert a["gn"] == b["gn"] and a["radius"] == b["radius"] and a["hist"] == b["hist"] and a["losses"] == b["losses"]


@pytest.mark.parametrize("policy", ["lstm64", "lstm256x2"])
def test_abi_collectives_single_rank_path(tmp_path, policy):
    """The same rehearsal on the C ABI's own communicator (include/uavppo.h K9: uav_comm_init, uav_allreduce, uav_allreduce_f64,
    uav_allgather_bytes -- RCCL bound by dlopen inside libuavppo.so, no torch.distributed group at all): one rank,
    UAVPPO_FORCE_COLLECTIVES=1, every exchange issued on the trainer's streams; BIT-identical to the job without collectives."""
    port = 29800 + os.getpid() % 1000
    _run(1, str(tmp_path / "plain"), port, policy)
    _run(1, str(tmp_path / "abi"), port + 1, policy, COLLECTIVES="abi", UAVPPO_FORCE_COLLECTIVES="1")
    a, b = torch.load(tmp_path / "plain.0"), torch.load(tmp_path / "abi.0")
    for k in ("flat", "adv", "obs"):
        assert torch.equal(a[k], b[k]), k
    assert a["gn"] == b["gn"] and a["radius"] == b["radius"] and a["hist"] == b["hist"] and a["losses"] == b["losses"]


def test_abi_collectives_refuse_without_a_communicator():
    from uavppo import ops
    t = torch.zeros(8, device="cuda:0")
    with pytest.raises(RuntimeError, match="no communicator"):
        ops.comm_allreduce(t)
    with pytest.raises(RuntimeError, match="no communicator"):
        ops.comm_allgather_bytes(torch.zeros(8, dtype=torch.uint8, device="cuda:0"))
    assert ops.comm_world("cuda:0") == 0


@pytest.mark.parametrize("mode", ["plain", "rccl-one-rank", "two-ranks-gloo"])
def test_bench_prints_exactly_one_json_line(mode):
    """The driver reads ONE JSON line from bench.py's stdout.  RCCL writes a version banner to stdout when its first
    communicator comes up, so bench.py sends everything but the line to stderr; checked with a real RCCL communicator
    (one rank) and through bench.py's own launcher (two ranks on this one GPU, gloo)."""
    import json
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if mode == "rccl-one-rank":
        env["UAVPPO_FORCE_COLLECTIVES"] = "1"
    if mode == "two-ranks-gloo":
        cmd += ["--gpus", "2", "--backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == (2 if mode == "two-ranks-gloo" else 1) and out["value"] > 0
    assert ("rehearsal" in out) == (mode == "rccl-one-rank")
    if mode == "two-ranks-gloo":
        assert out["strong_scaling"]["num_envs_total"] == 256
