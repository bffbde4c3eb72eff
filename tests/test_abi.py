"""CPU: the C-ABI library loads here (no GPU needed) and exports exactly what include/uavppo.h
declares; the ctypes table mirrors the header.  No compute call is made."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "uavppo.h")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uav_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from uavppo import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C uav-wrf-les-ppo-lstm_amd/csrc"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (uav_[a-z0-9_]+)", out))
    names = declared()
    assert len(names) >= 25
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in uavppo.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    from uavppo import _lib
    names = declared()
    assert sorted(_lib.SIGNATURES) == names
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n_decl = 0 if params in ("", "void") else len(params.split(","))
        assert n_decl == len(args), (name, n_decl, len(args))


def test_loads_and_reports_version_without_gpu():
    from uavppo import _lib
    lib = _lib.lib()
    assert lib.uav_abi_version() == 9
    assert lib.uav_mlp_param_count(6, 256, 128, 5) == 36230      # 36,230 parameters
    assert lib.uav_mlp_stash_floats(256, 128) == 2 * 256 + 2 * 128 + 2
    assert lib.uav_env_state_bytes(4096) > 4096 * 200


def test_collectives_bind_rccl_at_run_time_and_refuse_without_a_communicator():
    """comm.hip: libuavppo.so has NO link-time dependency on RCCL (it must load on a box without it); the version query binds
    librccl with dlopen, and the no-communicator answers need neither a GPU nor RCCL."""
    import ctypes as C
    from uavppo import _lib
    lib = _lib.lib()
    ldd = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in ldd and "nccl" not in ldd, ldd
    assert lib.uav_comm_world(None) == 0 and lib.uav_comm_rank(None) == -1
    v = C.c_int(0)
    rc = lib.uav_rccl_version(C.byref(v))
    if rc == 0:
        assert v.value > 20000                                    # e.g. 22606 = RCCL 2.26.6
    else:
        assert b"RCCL unavailable" in lib.uav_last_error()        # loud, with the reason


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a GPU-less host the first op raises instead of computing something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from uavppo import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.Context.get(torch.device("cuda", 0))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles the library (a no-op when it is current), the oracle's
    restatement, and asserts the ABI version the header declares."""
    import importlib
    import re
    g = importlib.import_module("__graft_entry__")
    g.build()
    header = open(os.path.join(ROOT, "include", "uavppo.h")).read()
    declared = int(re.search(r"#define UAV_ABI_VERSION (\d+)", header).group(1))
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert f"uav_abi_version() == {declared}" in src


def test_abi_carrier_needs_a_host_channel_for_the_unique_id():
    """dist_utils.use_abi_collectives: rank 0 draws the communicator's 128-byte id (host-only call) and needs a channel to hand it to
    the other ranks -- without an initialised torch.distributed group and without an explicit `uid` it must refuse, not guess."""
    import ctypes as C
    from uavppo import _lib, dist_utils, ops
    v = C.c_int(0)
    if _lib.lib().uav_rccl_version(C.byref(v)) != 0:
        pytest.skip("no RCCL on this host")
    uid = ops.comm_unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and uid != ops.comm_unique_id()      # a fresh id per draw
    with pytest.raises(RuntimeError, match="pass `uid`"):
        dist_utils.use_abi_collectives(0, 2)
    assert not dist_utils.abi_collectives() and dist_utils.world() == 1
    with pytest.raises(RuntimeError, match="128 bytes"):
        ops.comm_init(b"short", 0, 1)
