"""ctypes binding of libuavppo.so (the C ABI declared in include/uavppo.h).

There is NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised
(the reference's error convention, PPOV2.0/model.py:47-49).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch's bundled HIP runtime must be the one libuavppo.so binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UAVPPO_LIB") or os.path.join(_HERE, "libuavppo.so")   # override: instrumented builds

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
P, I32, I64, U64, F32, F64, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class EnvCfg(C.Structure):
    """struct uav_env_cfg (include/uavppo.h)."""
    _fields_ = [("variant", C.c_int32), ("field_mode", C.c_int32), ("n_fields", C.c_int32),
                ("bonus_is_f64", C.c_int32), ("env_offset", C.c_int32), ("n_env_total", C.c_int32),
                ("trend_k", C.c_int32), ("pad_", C.c_int32),
                ("radius", C.c_double), ("bonus", C.c_double),
                ("seed", C.c_uint64), ("bank", C.c_void_p), ("bank_src", C.c_void_p), ("curriculum", C.c_void_p)]


class StepperCall(C.Structure):
    """uav_stepper_call (include/uavppo.h)"""
    _fields_ = [("state", C.c_void_p), ("x", C.c_void_p), ("below", C.c_void_p), ("keep_t", C.c_void_p), ("t", C.c_int), ("I", C.c_int),
                ("y", C.c_void_p), ("stash", C.c_void_p), ("hn", C.c_void_p), ("cn", C.c_void_p)]


class LstmBwdLayer(C.Structure):
    """struct uav_lstm_bwd_layer (include/uavppo.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("keep", "stash", "w_hh", "w_ih", "dgates", "dx", "dhn", "dcn", "dh0", "dc0")]


# name -> (restype, argtypes); mirrors include/uavppo.h one to one
SIGNATURES = {
    "uav_abi_version": (I32, []),
    "uav_last_error": (C.c_char_p, []),
    "uav_create": (I32, [C.POINTER(C.c_void_p), I32, SZ]),
    "uav_destroy": (None, [P]),
    "uav_set_lstm_arith": (I32, [P, I32]),
    "uav_get_lstm_arith": (I32, [P]),
    "uav_set_debug_flags": (I32, [P, C.c_uint]),
    "uav_absmax": (I32, [P, P, I64, P, P]),
    "uav_gae": (I32, [P, P, P, P, P, I32, I32, F32, F32, I32, P, P]),
    "uav_adv_stats": (I32, [P, P, I64, P, P]),
    "uav_adv_normalise": (I32, [P, P, P, I64, P, P, P, P]),
    "uav_pack_success_bits": (I32, [P, P, I64, I32, P, P]),
    "uav_episode_rows": (I32, [P, P, P, P, I32, I32, I32, P, P, I32, P, P]),
    "uav_curriculum_state_bytes": (SZ, []),
    "uav_curriculum_init": (I32, [P, P, C.c_double, C.c_double, I32, P]),
    "uav_curriculum_update": (I32, [P, P, P, I32, I32, P]),
    "uav_ppo_loss": (I32, [P, P, P, P, P, P, P, P, I64, I32, F32, F32, F32, P, P, P, P, P]),
    "uav_ppo_loss_from_y": (I32, [P, P, P, P, P, P, P, P, P, I64, I32, I32, F32, F32, F32, P, P, P, P]),
    "uav_policy_sample": (I32, [P, P, I64, I32, P, U64, U64, I64, P, P, P, P, P, P]),
    "uav_policy_sample_at": (I32, [P, P, I64, I64, I32, U64, U64, I64, P, P, I32, I32, P, P, P, P, P]),
    "uav_store_transition": (I32, [P, I32, I32, I32, P, P, P, P, P, P, P, P, P]),
    "uav_rollout_tail": (I32, [P, P, I32, P, P, I64, I32, P, P, I32, P, I64, I32, I32, U64, U64, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "uav_clip_adam": (I32, [P, P, P, P, P, I64, I64, F32, F32, F32, F32, F32, P, P, P]),
    "uav_clip_adamw": (I32, [P, P, P, P, P, I64, I64, F32, F32, F32, F32, F32, F32, P, P]),
    "uav_smooth_l1": (I32, [P, P, P, I64, F32, P, P, P]),
    "uav_mse_bce": (I32, [P, P, P, I64, P, P, P]),
    "uav_lstm_stepper_bytes": (SZ, [I32, I32, I32]),
    "uav_lstm_stepper_begin": (I32, [P, P, P, P, P, P, P, P, I32, I32, I32, P]),
    "uav_lstm_stepper_step": (I32, [P, P, P, P, P, I32, I32, I32, I32, I32, P, P, P, P, P]),
    "uav_lstm_stepper_step_pair": (I32, [P, P, P, I32, I32, I32, P]),
    "uav_gemm_f32": (I32, [P, I64, I64, I64, P, I64, I64, P, I64, I64, P, I64, P, I32, P]),
    "uav_gemm_f16x3": (I32, [P, I64, I64, I64, P, I64, I64, P, I64, I64, P, I64, P, I32, P, P]),
    "uav_colsum": (I32, [P, P, I64, I32, P, P]),
    "uav_ln_relu": (I32, [P, P, P, P, P, P, I64, I32, P]),
    "uav_ln_relu_bwd": (I32, [P, P, P, P, P, P, I64, I32, P, P, P]),
    "uav_mlp_param_count": (I64, [I32, I32, I32, I32]),
    "uav_mlp_stash_floats": (I64, [I32, I32]),
    "uav_mlp_fwd": (I32, [P, P, P, I64, I32, I32, I32, I32, P, P, P]),
    "uav_mlp_bwd": (I32, [P, P, P, P, P, I64, I32, I32, I32, I32, P, P]),
    "uav_mlp_ppo_grad": (I32, [P, P, P, P, P, P, P, P, I64, I32, I32, I32, I32, F32, F32, F32, P, P, P]),
    "uav_lstm_fwd": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, P, P, P, P, P, P, I32, P, P]),
    "uav_lstm_bwd": (I32, [P, P, P, P, P, P, P, I32, P, P, I32, I32, I32, P, P, P, P, I32, P, P]),
    "uav_lstm_bwd_caps": (I32, [P, I32, I32]),
    "uav_lstm_dgates_bytes": (SZ, [P, I32, I32, I32]),
    "uav_lstm_dgates_f32": (I32, [P, P, I32, I32, I32, P, P]),
    "uav_lstm_bwd_stack": (I32, [P, I32, P, P, P, P, I32, I32, I32, I32, P]),
    "uav_lstm_wgrad": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P, P, P, P, P, P]),
    "uav_env_state_bytes": (SZ, [I32]),
    "uav_env_reset": (I32, [P, P, I32, C.POINTER(EnvCfg), P, P]),
    "uav_env_step": (I32, [P, P, I32, C.POINTER(EnvCfg), P, P, P, P, P, P, P, P, P, P]),
    "uav_env_peek": (I32, [P, P, I32, P, P, P, P, P]),
    "uav_env_materialise": (I32, [P, P, I32, C.POINTER(EnvCfg), I32, P, P]),
    "uav_rollout": (I32, [P, P, I32, C.POINTER(EnvCfg), I32, P, I32, I32, U64, P, P, P, P, P, P, P, P, P, P,
                          P, P, P, P, P, P, P, P, P, P]),
    "uav_rccl_version": (I32, [C.POINTER(C.c_int)]),
    "uav_comm_unique_id": (I32, [P]),
    "uav_comm_init": (I32, [P, P, I32, I32]),
    "uav_comm_world": (I32, [P]),
    "uav_comm_rank": (I32, [P]),
    "uav_comm_destroy": (I32, [P]),
    "uav_allreduce": (I32, [P, P, I64, P]),
    "uav_allreduce_f64": (I32, [P, P, I64, P]),
    "uav_allgather_bytes": (I32, [P, P, P, I64, P]),
}

_lib = None


def lib():
    """Load the shared library once; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C uav-wrf-les-ppo-lstm_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().uav_last_error().decode()}")
