"""Tensor-level wrappers over the C ABI: shape/dtype/device checks on the host, then one call
into libuavppo.so on torch's current HIP stream.  PyTorch only supplies device memory and
streams here; every number is produced by the hand-written HIP kernels."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import EnvCfg, check, lib

GAE_MODES = {"reference_exact": 0, "standard": 1}
WS_BYTES = 256 << 20


class Context:
    """One uav_ctx per device (opaque handle: CU count + scratch workspace)."""
    _per_device: dict = {}

    def __init__(self, device_index):
        if not torch.cuda.is_available():
            raise RuntimeError("uavppo needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        h = C.c_void_p()
        check(lib().uav_create(C.byref(h), int(device_index), WS_BYTES), "uav_create")
        self.handle = h
        self.device = device_index
        # the mode uav_create started the handle in (UAV_LSTM_F32_MFMA / UAV_LSTM_BF16X6 in the environment, read there once)
        self.initial_arith = {0: "fp16x3", 1: "bf16x6", 2: "f32_mfma"}[int(lib().uav_get_lstm_arith(h))]

    @classmethod
    def get(cls, device=None):
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        if idx not in cls._per_device:
            cls._per_device[idx] = cls(idx)
        return cls._per_device[idx]


def _h(t):
    return Context.get(t.device).handle


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None, shape=None, name="tensor"):
    """Device pointer of a checked tensor (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return C.c_void_p(t.data_ptr())


F32, F64, I32, U8 = torch.float32, torch.float64, torch.int32, torch.uint8


class _KernelTimer:
    """Optional HIP-event bracket around named launches on torch's current stream (bench.py uses it
    to time the dominant kernel live inside the timed region).  Disabled = zero overhead."""

    CAP = 48      # brackets kept per name: HIP hands timing events out of a pool of ~1 k; the allocation that grows it stalls
                  # the host for ~30 ms (seen as one slow iteration around the 37th of a run that bracketed every launch)

    def __init__(self):
        self.names, self.events = (), {}

    def enable(self, names):
        self.names, self.events = tuple(names), {n: [] for n in names}

    def disable(self):
        self.names, self.events = (), {}

    def bracket(self, name):
        if name not in self.names or len(self.events[name]) >= self.CAP:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events[name].append((a, b))
        a.record()
        return b

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, ev in self.events.items():
            if ev:
                ms = [a.elapsed_time(b) for a, b in ev]
                out[n] = {"avg_ms": sum(ms) / len(ms), "n": len(ms)}
        return out


KERNEL_TIMER = _KernelTimer()


# ----------------------------------------------------------------------------- G1 / G2
def gae(rew, val, done, gamma, lam, mode="reference_exact", last_val=None, out=None):
    n, T = rew.shape
    adv = torch.empty_like(rew) if out is None else out
    check(lib().uav_gae(_h(rew), _p(rew, F32, (n, T), "rew"), _p(val, F32, (n, T), "val"),
                        _p(done, F32, (n, T), "done"), _p(last_val, F32, (n,), "last_val"), n, T,
                        float(gamma), float(lam), GAE_MODES[mode], _p(adv, F32, (n, T), "adv"), _stream()),
          "uav_gae")
    return adv


def adv_stats(adv, out=None):
    stats = torch.empty(3, dtype=F64, device=adv.device) if out is None else out
    check(lib().uav_adv_stats(_h(adv), _p(adv, F32, name="adv"), adv.numel(), _p(stats, F64, (3,), "stats3"),
                              _stream()), "uav_adv_stats")
    return stats


def adv_normalise(adv, val, stats3, adv_out=None, ret_out=None):
    adv_out = torch.empty_like(adv) if adv_out is None else adv_out
    ret_out = torch.empty_like(adv) if ret_out is None else ret_out
    if val.shape != adv.shape:
        raise RuntimeError("adv_normalise: val/adv shape mismatch")
    check(lib().uav_adv_normalise(_h(adv), _p(adv, F32, name="adv"), _p(val, F32, name="val"), adv.numel(),
                                  _p(stats3, F64, (3,), "stats3"), _p(adv_out, F32, adv.shape, "adv_out"),
                                  _p(ret_out, F32, adv.shape, "ret_out"), _stream()), "uav_adv_normalise")
    return adv_out, ret_out


def pack_success_bits(flags, cap, out=None):
    """flags u8 [..] -> u8 [4 + cap + 1] message (count LE | success bits of the ended episodes in order | spare)."""
    f = flags.reshape(-1)
    out = torch.empty(4 + cap + 1, dtype=U8, device=f.device) if out is None else out
    check(lib().uav_pack_success_bits(_h(f), _p(f, U8, name="flags"), f.numel(), int(cap), _p(out, U8, (4 + cap + 1,), "msg"),
                                      _stream()), "uav_pack_success_bits")
    return out


def episode_rows(rew, info, flags, carry, rows, count, env_offset=0):
    """uav_episode_rows: per-episode sums of the episodes that ended in this rollout, appended to rows [cap, 12] f64 in arbitrary
    order (count i32 [1], zeroed by the caller); carry f64 [N, 8] holds the episodes in progress."""
    N, T = rew.shape
    check(lib().uav_episode_rows(_h(rew), _p(rew, F32, (N, T), "rew"), _p(info, F32, (N, T, 10), "info"), _p(flags, U8, (N, T), "flags"),
                                 N, T, int(env_offset), _p(carry, torch.float64, (N, 8), "carry"),
                                 _p(rows, torch.float64, (rows.shape[0], 12), "rows"), int(rows.shape[0]), _p(count, I32, (1,), "count"),
                                 _stream()), "uav_episode_rows")


def curriculum_state(device, radius=50.0, bonus=0.6, bonus_is_f64=False):
    """A device-side curriculum state (uav_curriculum_*): opaque u8 block, initialised."""
    st = torch.zeros(int(lib().uav_curriculum_state_bytes()), dtype=U8, device=device)
    curriculum_init(st, radius, bonus, bonus_is_f64)
    return st


def curriculum_init(state, radius, bonus, bonus_is_f64=False):
    check(lib().uav_curriculum_init(_h(state), _p(state, U8, name="curriculum state"), float(radius), float(bonus), int(bool(bonus_is_f64)),
                                    _stream()), "uav_curriculum_init")


def curriculum_update(state, msgs, cap):
    """Feed the success messages of all ranks (u8 [world, 4 + cap + 1], uav_pack_success_bits) to the device-side curriculum."""
    world = int(msgs.shape[0])
    check(lib().uav_curriculum_update(_h(state), _p(state, U8, name="curriculum state"), _p(msgs, U8, (world, 4 + cap + 1), "msgs"), world,
                                      int(cap), _stream()), "uav_curriculum_update")


def curriculum_read(state_host):
    """Decode a HOST copy (numpy u8 / torch cpu u8) of a curriculum state -> dict."""
    import numpy as np
    b = np.asarray(state_host, dtype=np.uint8)
    d = b[:32].view(np.float64)
    q = b[32:48].view(np.int64)
    i = b[48:56].view(np.int32)
    return {"radius": float(d[0]), "bonus": float(d[1]), "bonus_is_f64": bool(d[2] != 0.0), "overflow": bool(d[3] != 0.0),
            "episodes": int(q[0]), "successes": int(q[1]), "hist_len": int(i[0]), "win_succ": int(i[1])}


# ----------------------------------------------------------------------------- U2 / K3
def ppo_loss(logits, value, act, logp_old, adv, ret, val_old, inv_n, clip, ent_beta,
             loss_sums=None, dlogits=None, dvalue=None, dhead_bias=None):
    n, A = logits.shape
    loss_sums = torch.empty(4, dtype=F64, device=logits.device) if loss_sums is None else loss_sums
    dlogits = torch.empty_like(logits) if dlogits is None else dlogits
    dvalue = torch.empty(n, dtype=F32, device=logits.device) if dvalue is None else dvalue
    check(lib().uav_ppo_loss(_h(logits), _p(logits, F32, (n, A), "logits"), _p(value.reshape(-1), F32, (n,), "value"),
                             _p(act, I32, (n,), "act"), _p(logp_old, F32, (n,), "logp_old"),
                             _p(adv, F32, (n,), "adv"), _p(ret, F32, (n,), "ret"), _p(val_old, F32, (n,), "val_old"),
                             n, A, float(inv_n), float(clip), float(ent_beta), _p(loss_sums, F64, (4,), "loss_sums"),
                             _p(dlogits, F32, (n, A), "dlogits"), _p(dvalue, F32, (n,), "dvalue"),
                             _p(dhead_bias, F32, (A + 1,), "dhead_bias"), _stream()),
          "uav_ppo_loss")
    return loss_sums, dlogits, dvalue


def policy_sample(logits, u=None, seed=0, counter=0, forced_act=None, want_probs=False, nan_count=None, index_offset=0,
                  iteration=0):
    """Row i draws from Philox(seed; counter, index_offset + i, iteration) -- uav_rollout's key when counter = the step
    inside the rollout, index_offset = the global index of this rank's first env."""
    n, A = logits.shape
    dev = logits.device
    act = torch.empty(n, dtype=I32, device=dev)
    logp = torch.empty(n, dtype=F32, device=dev)
    probs = torch.empty(n, A, dtype=F32, device=dev) if want_probs else None
    nan_count = torch.zeros(1, dtype=I32, device=dev) if nan_count is None else nan_count
    check(lib().uav_policy_sample(_h(logits), _p(logits, F32, (n, A), "logits"), n, A, _p(u, F32, (n,), "u"),
                                  int(seed), (int(iteration) << 32) | (int(counter) & 0xffffffff), int(index_offset),
                                  _p(forced_act, I32, (n,), "forced_act"),
                                  _p(act), _p(logp), _p(probs), _p(nan_count, I32, (1,), "nan_count"), _stream()),
          "uav_policy_sample")
    return act, logp, probs, nan_count


def policy_sample_at(heads_seq, t, act_out, act_buf, val_buf, logp_buf, nan_count, seed=0, iteration=0, index_offset=0,
                     forced_act=None):
    """policy_sample for time step t of a step-wise rollout: logits | value read from heads_seq[:, t] ([N, T, A+1]), the
    action also to act_out [N] (the env step's input), action / value / log-prob to column t of the [N, T] buffers."""
    N, T, A1 = heads_seq.shape
    check(lib().uav_policy_sample_at(_h(heads_seq), C.c_void_p(_p(heads_seq, F32, (N, T, A1), "heads_seq").value + 4 * int(t) * A1),
                                     T * A1, N, A1 - 1, int(seed), (int(iteration) << 32) | (int(t) & 0xffffffff),
                                     int(index_offset), _p(forced_act, I32, (N,), "forced_act"), _p(act_out, I32, (N,), "act_out"),
                                     T, int(t), _p(act_buf, I32, (N, T), "act_buf"), _p(val_buf, F32, (N, T), "val_buf"),
                                     _p(logp_buf, F32, (N, T), "logp_buf"), _p(nan_count, I32, (1,), "nan_count"), _stream()),
          "uav_policy_sample_at")
    return act_out


def store_transition(t, keep, rew, done, flags, keep_buf, rew_buf, done_buf, flags_buf):
    """PPOBuffer.store's remaining columns for step t: keep / rew / done / flags [N] -> column t of the [N, T] buffers;
    keep becomes 1 - done (the next step's restart mask)."""
    N, T = keep_buf.shape
    check(lib().uav_store_transition(_h(keep), N, T, int(t), _p(keep, F32, (N,), "keep"), _p(rew, F32, (N,), "rew"),
                                     _p(done, F32, (N,), "done"), _p(flags, U8, (N,), "flags"), _p(keep_buf, F32, (N, T), "keep_buf"),
                                     _p(rew_buf, F32, (N, T), "rew_buf"), _p(done_buf, F32, (N, T), "done_buf"),
                                     _p(flags_buf, U8, (N, T), "flags_buf"), _stream()), "uav_store_transition")


def rollout_tail(env_state, cfg, y_seq, t, w_head, b_head, heads_seq, act_out, cur_obs, obs_seq, keep, act_buf, val_buf, logp_buf,
                 keep_buf, rew_buf, done_buf, flags_buf, nan_count, seed=0, iteration=0, index_offset=0, forced_act=None, noise=None):
    """uav_rollout_tail: everything of rollout step t after the recurrent layers in one launch -- heads_seq[:, t] from
    y_seq[:, t] ([N, T, H] top-layer output), the action draw, the environment step (auto-reset), PPOBuffer.store's columns
    (act / val / logp / keep / rew / done / flags [N, T] at t; keep [N] <- 1 - done) and the next observation into cur_obs and
    obs_seq[:, t + 1].  Identical results to gemm_rows + policy_sample_at + env_step + store_transition + the obs copy."""
    N, T, H = y_seq.shape
    A1 = heads_seq.shape[-1]
    od = cur_obs.shape[-1]
    check(lib().uav_rollout_tail(_h(y_seq), _p(env_state, U8, name="env state"), N, C.byref(cfg),
                                 C.c_void_p(_p(y_seq, F32, (N, T, H), "y_seq").value + 4 * int(t) * H), T * H, H,
                                 _p(w_head, F32, (A1, H), "w_head"), _p(b_head, F32, (A1,), "b_head"), A1 - 1,
                                 C.c_void_p(_p(heads_seq, F32, (N, T, A1), "heads_seq").value + 4 * int(t) * A1), T * A1, T, int(t),
                                 int(seed), (int(iteration) << 32) | (int(t) & 0xffffffff), int(index_offset),
                                 _p(forced_act, I32, (N,), "forced_act"), _p(noise, F64, (N, 2), "noise"),
                                 _p(act_out, I32, (N,), "act_out"), _p(cur_obs, F32, (N, od), "cur_obs"),
                                 _p(obs_seq, F32, (N, T, od), "obs_seq"), _p(keep, F32, (N,), "keep"),
                                 _p(act_buf, I32, (N, T), "act_buf"), _p(val_buf, F32, (N, T), "val_buf"),
                                 _p(logp_buf, F32, (N, T), "logp_buf"), _p(keep_buf, F32, (N, T), "keep_buf"),
                                 _p(rew_buf, F32, (N, T), "rew_buf"), _p(done_buf, F32, (N, T), "done_buf"),
                                 _p(flags_buf, U8, (N, T), "flags_buf"), _p(nan_count, I32, (1,), "nan_count"), _stream()),
          "uav_rollout_tail")
    return act_out


def ppo_loss_heads(heads, act, logp_old, adv, ret, val_old, inv_n, clip, ent_beta, loss_sums, dheads, dhead_bias=None):
    """Packed form of ppo_loss: heads [n, A+1] (logits | value) in, dheads [n, A+1] out -- no split/concat copies."""
    n, A1 = heads.shape
    check(lib().uav_ppo_loss(_h(heads), _p(heads, F32, (n, A1), "heads"), None, _p(act, I32, (n,), "act"),
                             _p(logp_old, F32, (n,), "logp_old"), _p(adv, F32, (n,), "adv"), _p(ret, F32, (n,), "ret"),
                             _p(val_old, F32, (n,), "val_old"), n, A1 - 1, float(inv_n), float(clip), float(ent_beta),
                             _p(loss_sums, F64, (4,), "loss_sums"), _p(dheads, F32, (n, A1), "dheads"), None,
                             _p(dhead_bias, F32, (A1,), "dhead_bias"), _stream()), "uav_ppo_loss")
    return loss_sums, dheads


def ppo_loss_from_y(y, w_head, b_head, act, logp_old, adv, ret, val_old, inv_n, clip, ent_beta, loss_sums, dheads,
                    dhead_bias=None):
    """Heads + loss fused (csrc/loss.hip: ppo_loss_from_y_kernel): y [n, H] -> dheads [n, A+1]."""
    n, H = y.shape
    A1 = w_head.shape[0]
    _t = KERNEL_TIMER.bracket("ppo_loss")
    check(lib().uav_ppo_loss_from_y(_h(y), _p(y, F32, (n, H), "y"), _p(w_head, F32, (A1, H), "w_head"),
                                    _p(b_head, F32, (A1,), "b_head"), _p(act, I32, (n,), "act"),
                                    _p(logp_old, F32, (n,), "logp_old"), _p(adv, F32, (n,), "adv"),
                                    _p(ret, F32, (n,), "ret"), _p(val_old, F32, (n,), "val_old"), n, H, A1 - 1,
                                    float(inv_n), float(clip), float(ent_beta), _p(loss_sums, F64, (4,), "loss_sums"),
                                    _p(dheads, F32, (n, A1), "dheads"), _p(dhead_bias, F32, (A1,), "dhead_bias"),
                                    _stream()), "uav_ppo_loss_from_y")
    if _t is not None:
        _t.record()
    return loss_sums, dheads


# ----------------------------------------------------------------------------- U3
ARITH = {"fp16x3": 0, "bf16x6": 1, "f32_mfma": 2}


def set_lstm_arith(mode, device=None):
    """Operand arithmetic of the LSTM sequence kernels on this device's handle (include/uavppo.h, UAV_ARITH_*)."""
    check(lib().uav_set_lstm_arith(Context.get(device).handle, ARITH[mode]), "uav_set_lstm_arith")


class lstm_arith:
    """`with ops.lstm_arith("f32_mfma"):` -- run a block on another arithmetic, then restore the handle's mode."""

    def __init__(self, mode, device=None):
        self.mode, self.device = mode, device

    def __enter__(self):
        self.prev = get_lstm_arith(self.device)
        set_lstm_arith(self.mode, self.device)

    def __exit__(self, *exc):
        set_lstm_arith(self.prev, self.device)


DEBUG_FLAGS = {"step_f32": 1, "x_f32": 2, "dg_f32": 4, "gemm_tn_off": 0x10}


def set_debug_flags(*names, device=None):
    """A/B switches of the h = 256 step path on this device's handle (include/uavppo.h, UAV_DEBUG_*); no names = none."""
    check(lib().uav_set_debug_flags(Context.get(device).handle, sum(DEBUG_FLAGS[n] for n in names)), "uav_set_debug_flags")


# ---- K9: RCCL behind the ABI (csrc/comm.hip; include/uavppo.h "K9")
def rccl_version():
    """RCCL's version code (e.g. 22706) as the library the ABI binds reports it."""
    out = C.c_int(0)
    check(lib().uav_rccl_version(C.byref(out)), "uav_rccl_version")
    return int(out.value)


def comm_unique_id():
    """128 opaque bytes drawn by rank 0 (ncclGetUniqueId); hand them to every rank's comm_init over any host channel."""
    buf = C.create_string_buffer(128)
    check(lib().uav_comm_unique_id(buf), "uav_comm_unique_id")
    return bytes(buf.raw)


def comm_init(uid, rank, world, device=None):
    """Join the job's communicator with this device's handle (blocks until all `world` ranks have called it)."""
    if len(uid) != 128:
        raise RuntimeError("comm_init: the unique id is 128 bytes")
    check(lib().uav_comm_init(Context.get(device).handle, C.c_char_p(uid), int(rank), int(world)), "uav_comm_init")


def comm_world(device=None):
    return int(lib().uav_comm_world(Context.get(device).handle))


def comm_destroy(device=None):
    check(lib().uav_comm_destroy(Context.get(device).handle), "uav_comm_destroy")


def comm_allreduce(t):
    """In-place sum over the ranks on the current stream: f32 (the flat gradient) or f64 (advantage statistics, loss sums)."""
    if t.dtype == torch.float32:
        check(lib().uav_allreduce(_h(t), _p(t, F32, name="t"), t.numel(), _stream()), "uav_allreduce")
    elif t.dtype == torch.float64:
        check(lib().uav_allreduce_f64(_h(t), _p(t, torch.float64, name="t"), t.numel(), _stream()), "uav_allreduce_f64")
    else:
        raise RuntimeError(f"comm_allreduce: f32 or f64, got {t.dtype}")
    return t


def comm_allgather_bytes(msg):
    """[world, len(msg)] u8: every rank's message, in rank order."""
    w = comm_world(msg.device)
    if w < 1:
        raise RuntimeError("comm_allgather_bytes: no communicator on this device's handle")
    out = torch.empty(w, msg.numel(), dtype=torch.uint8, device=msg.device)
    check(lib().uav_allgather_bytes(_h(msg), _p(msg, torch.uint8, name="msg"), _p(out, torch.uint8, name="out"), msg.numel(), _stream()),
          "uav_allgather_bytes")
    return out


def get_lstm_arith(device=None):
    m = int(lib().uav_get_lstm_arith(Context.get(device).handle))
    return {v: k for k, v in ARITH.items()}[m]


def absmax(x, out=None):
    """max |x| (NaN -> inf) as a 1-element device tensor; no host sync."""
    out = torch.empty(1, dtype=F32, device=x.device) if out is None else out
    check(lib().uav_absmax(_h(x), _p(x, F32, name="x"), x.numel(), _p(out, F32, (1,), "out"), _stream()), "uav_absmax")
    return out


def clip_adam(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=0.5,
              gnorm_out=None, pmax_out=None):
    n = param.numel()
    for t, nm in ((grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if t.numel() != n:
            raise RuntimeError(f"clip_adam: {nm} has {t.numel()} elements, param has {n}")
    check(lib().uav_clip_adam(_h(param), _p(param, F32, name="param"), _p(grad, F32, name="grad"),
                              _p(exp_avg, F32, name="exp_avg"), _p(exp_avg_sq, F32, name="exp_avg_sq"), n, int(step),
                              float(lr), float(beta1), float(beta2), float(eps), float(max_norm),
                              _p(gnorm_out, F32, (1,), "gnorm_out"), _p(pmax_out, F32, (1,), "pmax_out"), _stream()),
          "uav_clip_adam")


# ----------------------------------------------------------------------------- GEMM
def gemm_rows(a, b_t, bias, out):
    """out[n, :] = a[n, :] @ b_t.T + bias for ROW-STRIDED 2-D views a [N, K] and out [N, M] (unit stride along the last
    index, any row stride: one time step of an [N, T, K] array) and a contiguous b_t [M, K]: uav_gemm_f32 with lda / ldc."""
    N, K = a.shape
    M = b_t.shape[0]
    if a.stride(1) != 1 or out.stride(1) != 1 or tuple(out.shape) != (N, M) or b_t.shape[1] != K:
        raise RuntimeError("gemm_rows: a [N, K] and out [N, M] need unit stride along the last index; b_t is [M, K]")
    for t, nm in ((a, "a"), (out, "out")):
        if not t.is_cuda or t.dtype != F32:
            raise RuntimeError(f"gemm_rows: {nm}: expected a float32 GPU tensor")
    check(lib().uav_gemm_f32(_h(a), N, M, K, C.c_void_p(a.data_ptr()), a.stride(0), 1, _p(b_t, F32, name="b_t"), 1, K,
                             C.c_void_p(out.data_ptr()), out.stride(0), _p(bias, F32, (M,), "bias"), 0, _stream()),
          "uav_gemm_f32")
    return out


def gemm(a, b, trans_a=False, trans_b=False, bias=None, out=None, accumulate=False, split_fp16=False, a_absmax=None):
    """out[M,N] (+)= op(a) @ op(b) + bias with 2-D row-major tensors (op = optional transpose).  split_fp16: the same
    product as three fp16 piece products per f32 product (uav_gemm_f16x3: M % 128 == 0, N % 128 == 0, operands inside
    fp16's range; a_absmax = 1-element device tensor with max |a| block-scales a gradient-sized operand)."""
    if a.dim() != 2 or b.dim() != 2:
        raise RuntimeError("gemm: 2-D tensors expected")
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if K != K2:
        raise RuntimeError(f"gemm: inner dimensions differ ({K} vs {K2})")
    sa_m, sa_k = (1, a.shape[1]) if trans_a else (a.shape[1], 1)
    sb_k, sb_n = (1, b.shape[1]) if trans_b else (b.shape[1], 1)
    if out is None:
        if accumulate:
            raise RuntimeError("gemm: accumulate needs out")
        out = torch.empty(M, N, dtype=F32, device=a.device)
    if split_fp16:
        check(lib().uav_gemm_f16x3(_h(a), M, N, K, _p(a, F32, name="a"), sa_m, sa_k, _p(b, F32, name="b"), sb_k, sb_n,
                                   _p(out, F32, (M, N), "out"), N, _p(bias, F32, (N,), "bias"), int(accumulate),
                                   _p(a_absmax, F32, (1,), "a_absmax"), _stream()), "uav_gemm_f16x3")
        return out
    check(lib().uav_gemm_f32(_h(a), M, N, K, _p(a, F32, name="a"), sa_m, sa_k, _p(b, F32, name="b"), sb_k, sb_n,
                             _p(out, F32, (M, N), "out"), N, _p(bias, F32, (N,), "bias"), int(accumulate), _stream()),
          "uav_gemm_f32")
    return out


# ----------------------------------------------------------------------------- M2
def mlp_param_count(in_dim=6, h1=256, h2=128, n_act=5):
    return int(lib().uav_mlp_param_count(in_dim, h1, h2, n_act))


def mlp_fwd(params, x, in_dim=6, h1=256, h2=128, n_act=5, stash=None):
    B = x.shape[0]
    if params.numel() != mlp_param_count(in_dim, h1, h2, n_act):
        raise RuntimeError("mlp_fwd: flat parameter buffer has the wrong size")
    heads = torch.empty(B, n_act + 1, dtype=F32, device=x.device)
    per = int(lib().uav_mlp_stash_floats(h1, h2))
    if stash is None:
        stash = torch.empty(B * per, dtype=F32, device=x.device)
    elif stash.numel() < B * per:
        raise RuntimeError("mlp_fwd: stash too small")
    check(lib().uav_mlp_fwd(_h(x), _p(params, F32, name="params"), _p(x, F32, (B, in_dim), "x"), B, in_dim, h1, h2,
                            n_act, _p(heads), _p(stash, F32, name="stash"), _stream()), "uav_mlp_fwd")
    return heads, stash


def mlp_bwd(params, x, stash, dheads, in_dim=6, h1=256, h2=128, n_act=5, grad=None):
    B = x.shape[0]
    grad = torch.empty_like(params) if grad is None else grad
    check(lib().uav_mlp_bwd(_h(x), _p(params, F32, name="params"), _p(x, F32, (B, in_dim), "x"),
                            _p(stash, F32, name="stash"), _p(dheads, F32, (B, n_act + 1), "dheads"), B, in_dim, h1,
                            h2, n_act, _p(grad, F32, params.shape, "grad"), _stream()), "uav_mlp_bwd")
    return grad


def mlp_ppo_grad(params, obs, act, logp_old, adv, ret, val_old, inv_n, clip, ent_beta, loss_sums, grad, in_dim=6, h1=256,
                 h2=128, n_act=5):
    """Fused forward + clipped-PPO loss + backward of the reference's MLP policy (csrc/mlp_fused.hip): obs [n, 6] and the
    per-sample scalars in, flat gradient (layout of `params`) and loss_sums f64[4] out; nothing else touches HBM."""
    n = obs.shape[0]
    _t = KERNEL_TIMER.bracket("mlp_ppo_grad")
    check(lib().uav_mlp_ppo_grad(_h(obs), _p(params, F32, (mlp_param_count(in_dim, h1, h2, n_act),), "params"),
                                 _p(obs, F32, (n, in_dim), "obs"), _p(act, I32, (n,), "act"),
                                 _p(logp_old, F32, (n,), "logp_old"), _p(adv, F32, (n,), "adv"), _p(ret, F32, (n,), "ret"),
                                 _p(val_old, F32, (n,), "val_old"), n, in_dim, h1, h2, n_act, float(inv_n), float(clip),
                                 float(ent_beta), _p(loss_sums, F64, (4,), "loss_sums"), _p(grad, F32, params.shape, "grad"),
                                 _stream()), "uav_mlp_ppo_grad")
    if _t is not None:
        _t.record()
    return grad


# ----------------------------------------------------------------------------- E1-E5
ENV_VARIANTS = {"v2.0": 0, "v2.1": 1, "v1.1": 2}
ENV_MAX_STEPS = {"v2.0": 1000, "v2.1": 1000, "v1.1": 5000}


def env_state_bytes(n_env):
    return int(lib().uav_env_state_bytes(int(n_env)))


def make_env_cfg(variant, radius, bonus, seed=0, bank=None, bank_src=None, env_offset=0, n_env_total=0, trend_k=0, curriculum=None):
    """uav_env_cfg (host struct).  bonus: python float -> the reference's f32 expression,
    numpy.float64 -> its f64 expression (see csrc/env_core.h, environment.py:133)."""
    import numpy as np
    cfg = EnvCfg()
    cfg.variant = ENV_VARIANTS[variant]
    cfg.field_mode = 1 if bank is not None else 0
    cfg.n_fields = 0 if bank is None else int(bank.shape[0])
    cfg.bonus_is_f64 = int(isinstance(bonus, np.float64))
    cfg.env_offset = int(env_offset)
    cfg.n_env_total = int(n_env_total)
    cfg.trend_k = int(trend_k)
    cfg.radius = float(radius)
    cfg.bonus = float(bonus)
    cfg.seed = int(seed)
    cfg.curriculum = None if curriculum is None else _p(curriculum, U8, name="curriculum state").value
    if bank is not None:
        F_ = bank.shape[0]
        cfg.bank = _p(bank, F64, (F_, 500, 500, 2), "bank").value
        cfg.bank_src = _p(bank_src, F64, (F_, 2), "bank_src").value
    return cfg


def env_reset(state, n_env, cfg, obs_out):
    check(lib().uav_env_reset(_h(state), _p(state, U8, name="env state"), n_env, C.byref(cfg),
                              _p(obs_out, F32, (n_env, 6 + cfg.trend_k), "obs_out"), _stream()), "uav_env_reset")


def env_step(state, n_env, cfg, act, obs_out, rew, done, flags, noise=None, info=None, term_obs=None, rew64=None):
    check(lib().uav_env_step(_h(state), _p(state, U8, name="env state"), n_env, C.byref(cfg),
                             _p(act, I32, (n_env,), "act"), _p(noise, F64, (n_env, 2), "noise"),
                             _p(obs_out, F32, (n_env, 6 + cfg.trend_k), "obs_out"), _p(rew, F32, (n_env,), "rew"),
                             _p(done, F32, (n_env,), "done"), _p(flags, U8, (n_env,), "flags"),
                             _p(info, F32, (n_env, 5), "info"), _p(term_obs, F32, (n_env, 6 + cfg.trend_k), "term_obs"),
                             _p(rew64, F64, (n_env,), "rew64"), _stream()), "uav_env_step")


def env_peek(state, n_env, pos=None, source=None, steps=None, episode=None):
    check(lib().uav_env_peek(_h(state), _p(state, U8, name="env state"), n_env, _p(pos, F32, (n_env, 2), "pos"),
                             _p(source, F64, (n_env, 2), "source"), _p(steps, I32, (n_env,), "steps"),
                             _p(episode, I32, (n_env,), "episode"), _stream()), "uav_env_peek")


# ----------------------------------------------------------------------------- L1
def lstm_fwd(x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, stash=None, want_stash=True, y=None, w_head=None, b_head=None,
             heads=None):
    """One nn.LSTM layer over x[N,T,I] (env-major).  Returns y[N,T,H], hn, cn, stash[N,T,6H].
    With w_head [A1,H], b_head [A1] and a heads [N,T,A1] buffer the kernel also writes heads = y W_head^T + b_head."""
    N, T, I = x.shape
    H = w_hh.shape[1]
    dev = x.device
    y = torch.empty(N, T, H, dtype=F32, device=dev) if y is None else y
    hn = torch.empty(N, H, dtype=F32, device=dev)
    cn = torch.empty(N, H, dtype=F32, device=dev)
    if stash is None and (want_stash or I > 8 or H not in (64, 128)):     # the non-fused / generic paths work through the stash
        stash = torch.empty(N, T, 6 * H, dtype=F32, device=dev)
    _t = KERNEL_TIMER.bracket("lstm_fwd")
    check(lib().uav_lstm_fwd(_h(x), _p(x, F32, (N, T, I), "x"), _p(keep, F32, (N, T), "keep"),
                             _p(h0, F32, (N, H), "h0"), _p(c0, F32, (N, H), "c0"), _p(w_ih, F32, (4 * H, I), "w_ih"),
                             _p(w_hh, F32, (4 * H, H), "w_hh"), _p(b_ih, F32, (4 * H,), "b_ih"),
                             _p(b_hh, F32, (4 * H,), "b_hh"), N, T, I, H, _p(y, F32, (N, T, H), "y"), _p(hn), _p(cn),
                             _p(stash, F32, (N, T, 6 * H), "stash"),
                             _p(w_head, F32, None if w_head is None else (w_head.shape[0], H), "w_head"),
                             _p(b_head, F32, None if w_head is None else (w_head.shape[0],), "b_head"),
                             0 if w_head is None else int(w_head.shape[0]),
                             _p(heads, F32, None if w_head is None else (N, T, w_head.shape[0]), "heads"), _stream()),
          "uav_lstm_fwd")
    if _t is not None:
        _t.record()
    return y, hn, cn, stash


class LstmStepper:
    """uav_lstm_fwd one time step per call for one layer (H = 256, fp16-split arithmetic): weights split once by begin(),
    recurrent state kept on the device in `state`; step(t) reads x[:, t] and fills y[:, t] / stash[:, t] of the [N, T]
    arrays the BPTT reads; its `keep` argument restarts the state of the envs whose episode ended in the previous step."""

    def __init__(self, N, I, H, device):
        nbytes = lib().uav_lstm_stepper_bytes(int(N), int(I), int(H))
        if nbytes == 0:
            raise RuntimeError(f"uav_lstm_stepper: N={N} I={I} H={H} not supported (H = 256, I <= 256)")
        self.N, self.I, self.H = int(N), int(I), int(H)
        self.state = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.hn = torch.empty(N, H, dtype=F32, device=device)
        self.cn = torch.empty(N, H, dtype=F32, device=device)

    def begin(self, w_ih, w_hh, b_ih, b_hh, h0, c0):
        N, I, H = self.N, self.I, self.H
        check(lib().uav_lstm_stepper_begin(_h(self.state), _p(self.state), _p(w_ih, F32, (4 * H, I), "w_ih"),
                                           _p(w_hh, F32, (4 * H, H), "w_hh"), _p(b_ih, F32, (4 * H,), "b_ih"),
                                           _p(b_hh, F32, (4 * H,), "b_hh"), _p(h0, F32, (N, H), "h0"), _p(c0, F32, (N, H), "c0"),
                                           N, I, H, _stream()), "uav_lstm_stepper_begin")

    def step(self, x, t, y, stash, below=None, keep=None):
        """keep: [N] restart mask of this step (0 where the env's episode ended in the previous step) or None.
        below: the LstmStepper of the layer below, already stepped to t (x is then its y array: the kernel reads h_t from
        that stepper's piece planes instead of gathering and splitting f32 rows)."""
        N, I, H = self.N, self.I, self.H
        T = x.shape[1]
        check(lib().uav_lstm_stepper_step(_h(self.state), _p(self.state), _p(x, F32, (N, T, I), "x"),
                                          None if below is None else _p(below.state), _p(keep, F32, (N,), "keep"), N, T, int(t),
                                          I, H, _p(y, F32, (N, T, H), "y"), _p(stash, F32, (N, T, 6 * H), "stash"), _p(self.hn),
                                          _p(self.cn), _stream()), "uav_lstm_stepper_step")


def _stepper_call(sp, x, t, y, stash, below, keep):
    N, I, H = sp.N, sp.I, sp.H
    T = x.shape[1]
    c = _lib.StepperCall()
    c.state = _p(sp.state).value
    c.x = _p(x, F32, (N, T, I), "x").value
    c.below = None if below is None else _p(below.state).value
    c.keep_t = None if keep is None else _p(keep, F32, (N,), "keep").value
    c.t, c.I = int(t), I
    c.y = _p(y, F32, (N, T, H), "y").value
    c.stash = _p(stash, F32, (N, T, 6 * H), "stash").value
    c.hn, c.cn = _p(sp.hn).value, _p(sp.cn).value
    return c


def lstm_stepper_step_pair(a, b):
    """uav_lstm_stepper_step_pair: two INDEPENDENT stepper steps as one launch.  a, b: (stepper, x, t, y, stash, below, keep) as for
    LstmStepper.step -- e.g. layer 1's step t + 1 and layer 2's step t (below = layer 1's stepper)."""
    ca, cb = _stepper_call(*a), _stepper_call(*b)
    sp, x = a[0], a[1]
    check(lib().uav_lstm_stepper_step_pair(_h(sp.state), C.byref(ca), C.byref(cb), sp.N, x.shape[1], sp.H, _stream()),
          "uav_lstm_stepper_step_pair")


def lstm_dgates_bytes(N, T, H, device):
    """uav_lstm_dgates_bytes: size of the gate-gradient buffer uav_lstm_bwd / _bwd_stack hand to uav_lstm_wgrad under the device
    handle's CURRENT arithmetic mode and debug flags."""
    n = int(lib().uav_lstm_dgates_bytes(Context.get(torch.device(device)).handle, int(N), int(T), int(H)))
    if n <= 0:
        raise RuntimeError(f"uav_lstm_dgates_bytes({N}, {T}, {H}) = {n}")
    return n


def lstm_dgates(N, T, H, device):
    """A gate-gradient buffer for lstm_bwd / lstm_bwd_stack / lstm_wgrad: f32 rows [N, T, 4H] -- except at h = 256 on the fp16-split
    arithmetic, where the library keeps the BPTT's fp16 piece chunks + one scale per (env, step) instead of f32 rows and the
    buffer is an opaque flat f32 tensor (lstm_dgates_f32 converts)."""
    nbytes = lstm_dgates_bytes(N, T, H, device)
    if nbytes == N * T * 4 * H * 4:
        return torch.empty(N, T, 4 * H, dtype=F32, device=device)
    return torch.empty((nbytes + 3) // 4, dtype=F32, device=device)


def _pdg(dgates, N, T, H):
    need = lstm_dgates_bytes(N, T, H, dgates.device)
    if dgates.numel() * 4 < need:
        raise RuntimeError(f"dgates: {dgates.numel() * 4} bytes, uav_lstm_dgates_bytes({N}, {T}, {H}) = {need} (allocate with ops.lstm_dgates)")
    return _p(dgates, F32, None, "dgates")


def lstm_dgates_f32(dgates, N, T, H):
    """uav_lstm_dgates_f32: the gate gradients as f32 rows [N, T, 4H], whatever form the buffer holds them in."""
    out = torch.empty(N, T, 4 * H, dtype=F32, device=dgates.device)
    check(lib().uav_lstm_dgates_f32(_h(dgates), _pdg(dgates, N, T, H), N, T, H, _p(out), _stream()), "uav_lstm_dgates_f32")
    return out


def lstm_wgrad(x, keep, h0, y, stash, dgates, w_ih, dheads=None):
    """The weight-gradient pass alone, from given gate gradients: dW_ih, dW_hh, db (and dW_head = dheads^T y)."""
    N, T, I = x.shape
    H = y.shape[-1]
    dev = x.device
    nh = 0 if dheads is None else dheads.shape[-1]
    dw_ih = torch.empty(4 * H, I, dtype=F32, device=dev)
    dw_hh = torch.empty(4 * H, H, dtype=F32, device=dev)
    db = torch.empty(4 * H, dtype=F32, device=dev)
    dw_head = torch.empty(nh, H, dtype=F32, device=dev) if nh else None
    check(lib().uav_lstm_wgrad(_h(x), _p(x, F32, (N, T, I), "x"), _p(keep, F32, (N, T), "keep"), _p(h0, F32, (N, H), "h0"),
                               _p(y, F32, (N, T, H), "y"), _p(stash, F32, (N, T, 6 * H), "stash"),
                               _pdg(dgates, N, T, H), _p(w_ih, F32, (4 * H, I), "w_ih"),
                               _p(dheads, F32, (N, T, nh), "dheads"), nh, N, T, I, H,
                               _p(dw_ih, F32, (4 * H, I), "dw_ih"), _p(dw_hh, F32, (4 * H, H), "dw_hh"),
                               _p(db, F32, (4 * H,), "db"), None, _p(dw_head, F32, (nh, H), "dw_head"), _p(None), _stream()),
          "uav_lstm_wgrad")
    return {"dw_ih": dw_ih, "dw_hh": dw_hh, "db": db, "dw_head": dw_head}


def lstm_bwd_caps(device, I, H):
    """uav_lstm_bwd_caps bit mask for a layer of input width I, hidden size H on `device`: 1 = forms dx itself, 2 = takes
    dheads + w_head instead of dy."""
    return int(lib().uav_lstm_bwd_caps(Context.get(torch.device(device)).handle, int(I), int(H)))


def lstm_bwd_stack(layers, keep, dy=None, dheads=None, w_head=None):
    """uav_lstm_bwd_stack: the BPTTs of a stack of h = 256 layers as one pipelined call.  layers: top first, dicts with
    stash [N,T,6H], w_hh, w_ih (None for the last), dgates (ops.lstm_dgates; distinct per layer), dx [N,T,H] (None for the last)."""
    N, T, H6 = layers[0]["stash"].shape
    H = H6 // 6
    arr = (_lib.LstmBwdLayer * len(layers))()
    for a, d in zip(arr, layers):
        a.keep = None if keep is None else _p(keep, F32, (N, T), "keep").value
        a.stash = _p(d["stash"], F32, (N, T, 6 * H), "stash").value
        a.w_hh = _p(d["w_hh"], F32, (4 * H, H), "w_hh").value
        a.w_ih = None if d.get("w_ih") is None else _p(d["w_ih"], F32, (4 * H, H), "w_ih").value
        a.dgates = _pdg(d["dgates"], N, T, H).value
        a.dx = None if d.get("dx") is None else _p(d["dx"], F32, (N, T, H), "dx").value
        a.dhn = a.dcn = a.dh0 = a.dc0 = None
    nh = 0 if dheads is None else dheads.shape[-1]
    _t = KERNEL_TIMER.bracket("lstm_bwd")
    check(lib().uav_lstm_bwd_stack(_h(layers[0]["stash"]), len(layers), C.byref(arr), _p(dy, F32, (N, T, H), "dy"),
                                   _p(dheads, F32, (N, T, nh), "dheads"), _p(w_head, F32, (nh, H), "w_head"), nh, N, T, H, _stream()),
          "uav_lstm_bwd_stack")
    if _t is not None:
        _t.record()


def lstm_bwd(x, keep, stash, w_ih, w_hh, y, h0, dy=None, dheads=None, w_head=None, dhn=None, dcn=None, need_dx=False,
             dgates=None, dw_ih=None, dw_hh=None, db=None, dw_head=None, want_dstate=True, wgrad_dheads=None, bwd_done=False,
             db_hh=None):
    """BPTT sequence kernel + fused weight-gradient pass of one layer.  y, h0: the layer's forward
    output and initial hidden state (h_prev of the weight gradient is y shifted by one step).
    db_hh: a second [4H] tensor that receives the bias gradient too (nn.LSTM's bias_hh; saves the caller a copy launch).
    bwd_done: dgates (and the dx a layer above needs) were already produced by lstm_bwd_stack: only the weight gradients."""
    N, T, I = x.shape
    H = w_hh.shape[1]
    dev = x.device
    dgates = lstm_dgates(N, T, H, dev) if dgates is None else dgates
    dx = torch.empty(N, T, I, dtype=F32, device=dev) if (need_dx and not bwd_done) else None
    dw_ih = torch.empty(4 * H, I, dtype=F32, device=dev) if dw_ih is None else dw_ih
    dw_hh = torch.empty(4 * H, H, dtype=F32, device=dev) if dw_hh is None else dw_hh
    db = torch.empty(4 * H, dtype=F32, device=dev) if db is None else db
    dh0 = torch.empty(N, H, dtype=F32, device=dev) if want_dstate else None
    dc0 = torch.empty(N, H, dtype=F32, device=dev) if want_dstate else None
    nh = 0 if dheads is None else dheads.shape[-1]
    if wgrad_dheads is None:
        wgrad_dheads = dheads          # head-weight gradient wanted whenever dheads drives the backward
    nhw = 0 if wgrad_dheads is None else wgrad_dheads.shape[-1]
    if wgrad_dheads is not None and dw_head is None:
        dw_head = torch.empty(nhw, H, dtype=F32, device=dev)
    # the h = 256 step path forms dx = dG W_ih inside its per-step recurrent product (same dG fragments) when I == H
    dx_in_bwd = bool(need_dx and lstm_bwd_caps(x.device, I, H) & 1) or bwd_done
    _t = None if bwd_done else KERNEL_TIMER.bracket("lstm_bwd")
    if not bwd_done:
        check(lib().uav_lstm_bwd(_h(x), _p(keep, F32, (N, T), "keep"), _p(stash, F32, (N, T, 6 * H), "stash"),
                                 _p(w_hh, F32, (4 * H, H), "w_hh"), _p(dy, F32, (N, T, H), "dy"),
                                 _p(dheads, F32, (N, T, nh), "dheads"), _p(w_head, F32, (nh, H), "w_head"), nh,
                                 _p(dhn, F32, (N, H), "dhn"), _p(dcn, F32, (N, H), "dcn"), N, T, H,
                                 _pdg(dgates, N, T, H), _p(dh0), _p(dc0),
                                 _p(w_ih, F32, (4 * H, I), "w_ih") if dx_in_bwd else None, I, _p(dx) if dx_in_bwd else None,
                                 _stream()), "uav_lstm_bwd")
    if _t is not None:
        _t.record()
    _t = KERNEL_TIMER.bracket("lstm_wgrad")
    check(lib().uav_lstm_wgrad(_h(x), _p(x, F32, (N, T, I), "x"), _p(keep, F32, (N, T), "keep"), _p(h0, F32, (N, H), "h0"),
                               _p(y, F32, (N, T, H), "y"), _p(stash, F32, (N, T, 6 * H), "stash"),
                               _pdg(dgates, N, T, H), _p(w_ih, F32, (4 * H, I), "w_ih"),
                               _p(wgrad_dheads, F32, (N, T, nhw), "dheads"), nhw, N, T, I, H,
                               _p(dw_ih, F32, (4 * H, I), "dw_ih"), _p(dw_hh, F32, (4 * H, H), "dw_hh"),
                               _p(db, F32, (4 * H,), "db"), _p(db_hh, F32, (4 * H,), "db_hh"), _p(dw_head, F32, (nhw, H), "dw_head"),
                               None if dx_in_bwd else _p(dx), _stream()),
          "uav_lstm_wgrad")
    if _t is not None:
        _t.record()
    return {"dx": dx, "dw_ih": dw_ih, "dw_hh": dw_hh, "db": db, "dh0": dh0, "dc0": dc0, "dgates": dgates,
            "dw_head": dw_head}


def env_materialise(state, n_env, cfg, env_index, out=None):
    out = torch.empty(500, 500, 2, dtype=F64, device=state.device) if out is None else out
    check(lib().uav_env_materialise(_h(state), _p(state, U8, name="env state"), n_env, C.byref(cfg), int(env_index),
                                    _p(out, F64, (500, 500, 2), "field_out"), _stream()), "uav_env_materialise")
    return out


def ln_relu(z, gamma, beta, want_stats=False):
    """relu(LayerNorm(z)) over the rows of z [rows, cols]; z is overwritten with the normalised values.
    want_stats: also return rstd [rows] (with z = xhat, what ln_relu_bwd needs)."""
    rows, cols = z.shape
    a = torch.empty_like(z)
    rstd = torch.empty(rows, dtype=F32, device=z.device)
    check(lib().uav_ln_relu(_h(z), _p(z, F32, (rows, cols), "z"), _p(a, F32), _p(rstd, F32), _p(gamma, F32, (cols,), "gamma"),
                            _p(beta, F32, (cols,), "beta"), rows, cols, _stream()), "uav_ln_relu")
    return (a, rstd) if want_stats else a


def ln_relu_bwd(d, xhat, rstd, gamma, beta):
    """d [rows, cols] = dL/da on entry, dL/dz on return (in place); returns (d, dgamma, dbeta)."""
    rows, cols = d.shape
    dg = torch.empty(cols, dtype=F32, device=d.device)
    db = torch.empty(cols, dtype=F32, device=d.device)
    check(lib().uav_ln_relu_bwd(_h(d), _p(d, F32, (rows, cols), "d"), _p(xhat, F32, (rows, cols), "xhat"),
                                _p(rstd, F32, (rows,), "rstd"), _p(gamma, F32, (cols,), "gamma"), _p(beta, F32, (cols,), "beta"),
                                rows, cols, _p(dg, F32), _p(db, F32), _stream()), "uav_ln_relu_bwd")
    return d, dg, db


def smooth_l1(pred, target, beta):
    """nn.SmoothL1Loss(beta) (mean) forward + backward: returns (loss f64[1] device tensor, dpred [n])."""
    n = pred.numel()
    loss = torch.zeros(1, dtype=F64, device=pred.device)
    dpred = torch.empty(n, dtype=F32, device=pred.device)
    check(lib().uav_smooth_l1(_h(pred), _p(pred, F32, (n,), "pred"), _p(target, F32, (n,), "target"), n, float(beta),
                              _p(loss, F64), _p(dpred, F32), _stream()), "uav_smooth_l1")
    return loss, dpred


def mse_bce(out, target):
    """MSE on column 0 + BCE(sigmoid(column 1)) (PPOV2.1/train_lstm.py:110-113): returns (loss f64[1], dout [n, 2])."""
    n = out.shape[0]
    loss = torch.zeros(1, dtype=F64, device=out.device)
    dout = torch.empty(n, 2, dtype=F32, device=out.device)
    check(lib().uav_mse_bce(_h(out), _p(out, F32, (n, 2), "out"), _p(target, F32, (n, 2), "target"), n, _p(loss, F64),
                            _p(dout, F32), _stream()), "uav_mse_bce")
    return loss, dout


def clip_adamw(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, max_norm=1.0,
               gnorm_out=None):
    """clip_grad_norm_(max_norm) + torch.optim.AdamW step on flat buffers (train_lstm.py:67,91-92)."""
    n = param.numel()
    check(lib().uav_clip_adamw(_h(param), _p(param, F32, (n,), "param"), _p(grad, F32, (n,), "grad"),
                               _p(exp_avg, F32, (n,), "exp_avg"), _p(exp_avg_sq, F32, (n,), "exp_avg_sq"), n, int(step),
                               float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), float(max_norm),
                               _p(gnorm_out, F32, (1,), "gnorm_out"), _stream()), "uav_clip_adamw")


def colsum(x, out=None):
    rows, cols = x.shape
    out = torch.empty(cols, dtype=F32, device=x.device) if out is None else out
    check(lib().uav_colsum(_h(x), _p(x, F32, (rows, cols), "x"), rows, cols, _p(out, F32, (cols,), "out"), _stream()),
          "uav_colsum")
    return out


# ----------------------------------------------------------------------------- R1
def rollout_lstm(env_state, n_env, cfg, params, hidden, horizon, it, cur_obs, h, c, bufs, last_val=None,
                 forced_act=None, noise=None, nan_count=None, stash=None, y=None, info=None, heads=None):
    """Fused persistent rollout (csrc/rollout.hip).  bufs: dict obs[N,T,6] act rew val logp done flags keep."""
    N, T = n_env, horizon
    _t = KERNEL_TIMER.bracket("rollout")
    check(lib().uav_rollout(_h(cur_obs), _p(env_state, U8, name="env state"), N, C.byref(cfg), 1,
                            _p(params, F32, name="params"), int(hidden), T, int(it),
                            _p(cur_obs, F32, (N, 6), "cur_obs"), _p(h, F32, (N, hidden), "h"), _p(c, F32, (N, hidden), "c"),
                            _p(bufs["obs"], F32, (N, T, 6), "obs"), _p(bufs["act"], I32, (N, T), "act"),
                            _p(bufs["rew"], F32, (N, T), "rew"), _p(bufs["val"], F32, (N, T), "val"),
                            _p(bufs["logp"], F32, (N, T), "logp"), _p(bufs["done"], F32, (N, T), "done"),
                            _p(bufs["flags"], U8, (N, T), "flags"), _p(bufs["keep"], F32, (N, T), "keep"),
                            _p(last_val, F32, (N,), "last_val"), _p(forced_act, I32, (N, T), "forced_act"),
                            _p(noise, F64, (N, T, 2), "noise"), _p(nan_count, I32, (1,), "nan_count"),
                            _p(stash, F32, (N, T, 6 * hidden), "stash"), _p(y, F32, (N, T, hidden), "y"),
                            _p(info, F32, (N, T, 10), "info"), _p(heads, F32, (N, T, 6), "heads"), _stream()),
          "uav_rollout")
    if _t is not None:
        _t.record()


def rollout_mlp(env_state, n_env, cfg, params, horizon, it, cur_obs, bufs, last_val=None, forced_act=None, noise=None,
                nan_count=None, info=None, heads=None):
    """Fused persistent rollout of the reference's MLP policy (csrc/mlp_fused.hip): uav_rollout with policy_kind 0."""
    N, T = n_env, horizon
    _t = KERNEL_TIMER.bracket("rollout")
    check(lib().uav_rollout(_h(cur_obs), _p(env_state, U8, name="env state"), N, C.byref(cfg), 0,
                            _p(params, F32, name="params"), 0, T, int(it), _p(cur_obs, F32, (N, 6), "cur_obs"), None, None,
                            _p(bufs["obs"], F32, (N, T, 6), "obs"), _p(bufs["act"], I32, (N, T), "act"),
                            _p(bufs["rew"], F32, (N, T), "rew"), _p(bufs["val"], F32, (N, T), "val"),
                            _p(bufs["logp"], F32, (N, T), "logp"), _p(bufs["done"], F32, (N, T), "done"),
                            _p(bufs["flags"], U8, (N, T), "flags"), None, _p(last_val, F32, (N,), "last_val"),
                            _p(forced_act, I32, (N, T), "forced_act"), _p(noise, F64, (N, T, 2), "noise"),
                            _p(nan_count, I32, (1,), "nan_count"), None, None, _p(info, F32, (N, T, 10), "info"),
                            _p(heads, F32, (N, T, 6), "heads"), _stream()), "uav_rollout")
    if _t is not None:
        _t.record()
