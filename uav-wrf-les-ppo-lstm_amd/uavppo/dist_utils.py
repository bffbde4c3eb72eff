"""Data-parallel plumbing: env shards per rank, and the three exchanges of one iteration
(SURVEY 8e).  Backend-agnostic torch.distributed calls ('nccl' == RCCL over xGMI on the GPU box;
'gloo' in the CPU tests).  No data-path collective exists: rollout buffers never leave a rank."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def env_shard(rank, envs_per_rank):
    """Global env indices [lo, hi) owned by `rank` (contiguous shards; RNG keys and the
    materialised-field choice use the GLOBAL index, so a job is invariant to how it is sharded)."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def allreduce_adv_stats(stats3):
    """(sum, sum of squares, count) of the advantages over ALL ranks: the reference normalises over
    the whole buffer (train_ppo2.0.py:35-39)."""
    if world() > 1:
        dist.all_reduce(stats3)
    return stats3


def allreduce_grad(flat_grad):
    """ONE all-reduce (sum) of the flat gradient per optimiser step.  Every rank's loss is already
    scaled by 1/(global sample count), so the sum IS the global-mean gradient; the clip norm is
    computed after it, identically on all ranks."""
    if world() > 1:
        dist.all_reduce(flat_grad)
    return flat_grad


def gather_episode_successes(flags):
    """Success bits of the episodes that ENDED in this rollout, over all ranks, in (rank, env, time) =
    global (env, time) order, as a host bool array.  Only the compacted bits travel: flags is non-zero
    exactly where an episode ended (bit0 done, bit1 reached), so one device-side nonzero() + two tiny
    all-gathers (counts, then padded bits) replace shipping the whole [N, T] array to every host."""
    f = flags.reshape(-1)
    idx = torch.nonzero(f).squeeze(1)
    succ = ((f[idx] >> 1) & 1).to(torch.uint8)
    if world() > 1:
        cnt = torch.tensor([succ.numel()], dtype=torch.int64, device=f.device)
        cnts = [torch.zeros_like(cnt) for _ in range(world())]
        dist.all_gather(cnts, cnt)
        cnts = [int(c.item()) for c in cnts]
        m = max(max(cnts), 1)
        pad = torch.zeros(m, dtype=torch.uint8, device=f.device)
        pad[:succ.numel()] = succ
        parts = [torch.empty_like(pad) for _ in range(world())]
        dist.all_gather(parts, pad)
        succ = torch.cat([p[:c] for p, c in zip(parts, cnts)])
    return succ.cpu().numpy().astype(bool)


def gather_episode_flags(flags):
    """[N_local, T] u8 flags of every rank concatenated in rank (= global env) order, so all ranks
    feed the SAME episode sequence to their replicated curriculum."""
    if world() == 1:
        return flags
    parts = [torch.empty_like(flags) for _ in range(world())]
    dist.all_gather(parts, flags)
    return torch.cat(parts, 0)
