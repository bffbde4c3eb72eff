"""CPU, world_size 2 and 8 (the size of the driver's first real multi-GPU run), gloo: the data-parallel exchanges of one iteration (uavppo/dist_utils.py)
give the same numbers as one process over the whole buffer.  Compute on each rank is the oracle
(no GPU here); what is under test is the sharding + collective logic the GPU trainer uses."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ppo_oracle as po


def _data():
    rng = np.random.RandomState(0)
    N, T = 16, 16
    obs = rng.rand(N * T, 6).astype(np.float32)
    act = rng.randint(0, 5, N * T)
    adv = (rng.randn(N, T) * 2 + 0.5).astype(np.float32)
    val = rng.randn(N, T).astype(np.float32)
    logp = (np.log(0.2) + 0.1 * rng.randn(N * T)).astype(np.float32)
    flags = (rng.rand(N, T) < 0.1).astype(np.uint8) * np.where(rng.rand(N, T) < 0.5, 3, 1).astype(np.uint8)
    torch.manual_seed(0)
    p = {"feature.0.weight": torch.randn(256, 6) * 0.3, "feature.0.bias": torch.zeros(256),
         "feature.1.weight": torch.ones(256), "feature.1.bias": torch.zeros(256),
         "feature.3.weight": torch.randn(128, 256) * 0.1, "feature.3.bias": torch.zeros(128),
         "feature.4.weight": torch.ones(128), "feature.4.bias": torch.zeros(128),
         "actor.weight": torch.randn(5, 128) * 0.1, "actor.bias": torch.zeros(5),
         "critic.weight": torch.randn(1, 128) * 0.1, "critic.bias": torch.zeros(1)}
    return N, T, obs, act, adv, val, logp, flags, p


def _grad(p, obs, act, logp, adv_n, ret, val, inv_n):
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    probs, value, _ = po.mlp_forward(leaf, torch.from_numpy(obs))
    total, _, _, _ = po.ppo_losses(probs, value, torch.from_numpy(act), torch.from_numpy(logp), adv_n, ret,
                                   torch.from_numpy(val))
    (total * (len(obs) * inv_n)).backward()          # mean over the shard -> sum/global count
    return torch.cat([leaf[k].grad.reshape(-1) for k in po.MLP_KEYS])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uavppo import dist_utils as du
    N, T, obs, act, adv, val, logp, flags, p = _data()
    lo, hi = du.env_shard(rank, N // world)
    a = adv[lo:hi].astype(np.float64)
    stats = torch.tensor([a.sum(), (a * a).sum(), a.size], dtype=torch.float64)
    du.allreduce_adv_stats(stats)
    cnt = stats[2].item()
    mean = stats[0].item() / cnt
    std = np.sqrt((stats[1].item() - cnt * mean * mean) / (cnt - 1))
    adv_n = torch.from_numpy(((adv[lo:hi] - np.float32(mean)) / (np.float32(std) + np.float32(1e-6))).reshape(-1))
    ret = adv_n + torch.from_numpy(val[lo:hi].reshape(-1))
    sl = slice(lo * T, hi * T)
    g = _grad(p, obs[sl], act[sl], logp[sl], adv_n, ret, val[lo:hi].reshape(-1), 1.0 / (N * T))
    du.allreduce_grad(g)
    fl = du.gather_episode_flags(torch.from_numpy(flags[lo:hi].copy()))
    succ = du.gather_episode_successes(torch.from_numpy(flags[lo:hi].copy()))
    cap = du.SUCC_CAP
    du.SUCC_CAP = 3                                   # force the overflow fallback (whole flags arrays)
    succ_overflow = du.gather_episode_successes(torch.from_numpy(flags[lo:hi].copy()))
    du.SUCC_CAP = cap
    if rank == 0:
        torch.save({"adv_n": adv_n, "grad": g, "flags": fl, "cnt": cnt, "succ": torch.from_numpy(succ),
                    "succ_overflow": torch.from_numpy(succ_overflow)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_exchange_equals_single_process(tmp_path, world):
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    N, T, obs, act, adv, val, logp, flags, p = _data()
    adv_n, ret = po.normalise(adv, val)                     # single process over the whole buffer
    assert got["cnt"] == N * T
    assert torch.allclose(got["adv_n"], adv_n[: N * T // world], atol=1e-6)
    g = _grad(p, obs, act, logp, adv_n, ret, val.reshape(-1), 1.0 / (N * T))
    assert torch.allclose(got["grad"], g, rtol=1e-4, atol=1e-7)
    assert np.array_equal(got["flags"].numpy(), flags)
    ended = (flags & 1) > 0
    assert np.array_equal(got["succ"].numpy(), ((flags & 2) > 0)[ended])        # global (env, time) order
    assert np.array_equal(got["succ_overflow"].numpy(), ((flags & 2) > 0)[ended])
