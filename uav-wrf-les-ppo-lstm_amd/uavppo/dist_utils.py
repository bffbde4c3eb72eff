"""Data-parallel plumbing: env shards per rank, and the three exchanges of one iteration
(SURVEY 8e).  Two carriers, the same three calls:

  torch.distributed (default)   backend-agnostic ('nccl' == RCCL over xGMI on the GPU box; 'gloo' in the CPU tests)
  the C ABI (UAVPPO_COLLECTIVES=abi, or use_abi_collectives())   uav_allreduce / uav_allreduce_f64 / uav_allgather_bytes of
                                include/uavppo.h on the handle's own RCCL communicator, issued on the caller's stream -- what
                                a host that is not PyTorch would call (INTEGRATION.md "Collectives")

No data-path collective exists: rollout buffers never leave a rank."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

_ABI = {"on": False, "world": 1, "rank": 0}


def use_abi_collectives(rank, world_size, device=None, uid=None):
    """Carry the iteration's exchanges on the C ABI's own RCCL communicator.  Rank 0 draws the 128-byte id; with an initialised
    torch.distributed group (any backend: it is only the host channel for those bytes) it is broadcast through it, otherwise
    the caller passes `uid` itself (a file, MPI, a socket ...).  Call before the trainer is built, like init_process_group."""
    from . import ops
    if uid is None:
        box = [ops.comm_unique_id() if rank == 0 else None]
        if world_size > 1:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("use_abi_collectives: pass `uid` (rank 0's ops.comm_unique_id()) or initialise torch.distributed "
                                   "as the host channel for it")
            dist.broadcast_object_list(box, src=0)
        uid = box[0]
    ops.comm_init(uid, rank, world_size, device)
    _ABI.update(on=True, world=int(world_size), rank=int(rank))


def abi_collectives():
    return _ABI["on"]


def world():
    if _ABI["on"]:
        return _ABI["world"]
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    """True when the iteration's exchanges must be issued: more than one rank, or UAVPPO_FORCE_COLLECTIVES=1 with an
    initialised process group / ABI communicator (a one-rank RCCL communicator then carries every exchange unchanged: how the
    RCCL code path is exercised on a one-GPU box, tests/test_gpu_multirank.py::test_rccl_single_rank_path)."""
    forced = os.environ.get("UAVPPO_FORCE_COLLECTIVES") == "1"
    if _ABI["on"]:
        return _ABI["world"] > 1 or forced
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or forced


def allreduce_sum(t):
    """In-place sum of a small f32 / f64 tensor over the ranks on whichever carrier is up (no-op without collectives)."""
    if collectives_on():
        if _ABI["on"] and t.is_cuda:
            from . import ops
            ops.comm_allreduce(t)
        else:
            dist.all_reduce(t)
    return t


def env_shard(rank, envs_per_rank):
    """Global env indices [lo, hi) owned by `rank` (contiguous shards; RNG keys and the
    materialised-field choice use the GLOBAL index, so a job is invariant to how it is sharded)."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def allreduce_adv_stats(stats3):
    """(sum, sum of squares, count) of the advantages over ALL ranks: the reference normalises over
    the whole buffer (train_ppo2.0.py:35-39)."""
    return allreduce_sum(stats3)


def allreduce_grad(flat_grad):
    """ONE all-reduce (sum) of the flat gradient per optimiser step.  Every rank's loss is already
    scaled by 1/(global sample count), so the sum IS the global-mean gradient; the clip norm is
    computed after it, identically on all ranks."""
    return allreduce_sum(flat_grad)


SUCC_CAP = 16384          # per-rank capacity of the fixed-size success message (episodes ended in one rollout)


def pack_local_successes(flags):
    """Device side of the success exchange, no host sync: this rank's [count (4 bytes LE) | success bits of the ended
    episodes in (env, time) order, zero padded to SUCC_CAP | spare byte] as ONE fixed-size u8 message.  flags is non-zero
    exactly where an episode ended (bit0 done, bit1 reached); the position of an ended episode in the message is its
    exclusive prefix count: fixed shapes, one kernel launch on the GPU."""
    f = flags.reshape(-1)
    if f.is_cuda:                              # one launch (csrc/gae.hip: pack_success_kernel)
        from . import ops
        return ops.pack_success_bits(f, SUCC_CAP)
    ended = f != 0                             # CPU tensors (gloo tests): the same message with torch ops
    pos = torch.cumsum(ended.to(torch.int32), 0) - 1
    cnt = ended.sum().to(torch.int32)
    msg = torch.zeros(4 + SUCC_CAP + 1, dtype=torch.uint8, device=f.device)     # last byte: dump slot of the scatter
    msg[:4] = torch.stack([(cnt >> s) & 255 for s in (0, 8, 16, 24)]).to(torch.uint8)
    slot = torch.where(ended & (pos < SUCC_CAP), pos, torch.full_like(pos, SUCC_CAP)).to(torch.int64)
    msg[4:].scatter_(0, slot, ((f >> 1) & 1).to(torch.uint8))
    return msg


def exchange_successes(msg):
    """The messages of all ranks, stacked [world, 4 + SUCC_CAP + 1], after ONE all-gather (a view of `msg` at world 1)."""
    if not collectives_on():
        return msg[None]
    if _ABI["on"] and msg.is_cuda:
        from . import ops
        return ops.comm_allgather_bytes(msg)
    parts = [torch.empty_like(msg) for _ in range(world())]
    dist.all_gather(parts, msg)
    return torch.stack(parts)


def pack_episode_successes(flags):
    return exchange_successes(pack_local_successes(flags))


def unpack_episode_successes(host, flags):
    """Host side: the stacked messages (numpy u8 [world, 4 + SUCC_CAP + 1]) -> success bits of all ranks in global (env, time)
    order as a bool array.  A rank with more than SUCC_CAP ended episodes (rare) makes every rank fall back to gathering
    the whole flags arrays (`flags`: this rank's device tensor)."""
    import numpy as np
    counts = [int(h[0]) | int(h[1]) << 8 | int(h[2]) << 16 | int(h[3]) << 24 for h in host]
    if max(counts) > SUCC_CAP:
        allf = gather_episode_flags(flags).reshape(-1)
        idx = torch.nonzero(allf).squeeze(1)
        return ((allf[idx] >> 1) & 1).to(torch.uint8).cpu().numpy().astype(bool)
    return np.concatenate([h[4:4 + c] for h, c in zip(host, counts)]).astype(bool)


def gather_episode_successes(flags):
    """Success bits of the episodes that ENDED in this rollout, over all ranks, in (rank, env, time) = global (env, time)
    order, as a host bool array: pack (one all-gather) + one device-to-host copy + unpack.  The trainer runs the two
    halves apart -- pack on a side stream right behind the rollout, unpack when the curriculum needs it -- so the copy's
    host sync never idles the GPU; this is the same thing in one call."""
    return unpack_episode_successes(pack_episode_successes(flags).cpu().numpy(), flags)


def gather_episode_flags(flags):
    """[N_local, T] u8 flags of every rank concatenated in rank (= global env) order, so all ranks
    feed the SAME episode sequence to their replicated curriculum."""
    if not collectives_on():
        return flags
    if _ABI["on"] and flags.is_cuda:
        from . import ops
        return ops.comm_allgather_bytes(flags.reshape(-1)).reshape(world() * flags.shape[0], *flags.shape[1:])
    parts = [torch.empty_like(flags) for _ in range(world())]
    dist.all_gather(parts, flags)
    return torch.cat(parts, 0)
