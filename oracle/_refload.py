"""Import the reference's hot-path modules in the BUILD container (never on the GPU box).

Only `oracle/gen_golden.py` uses this.  `gym` and `netCDF4` are absent from the image and
contribute no arithmetic (base class + space objects / an output writer), so they are
replaced by empty stand-ins; the reference's own code runs untouched from /root/reference.
"""
import importlib.util
import sys
import types

REF_ROOT = "/root/reference"
_MODS = ("config", "environment", "model", "netcdf_writer", "train_ref")


def _install_third_party_stubs():
    gym = types.ModuleType("gym")

    class Env:
        def __init__(self):
            pass

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.dtype = low, high, dtype

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete, spaces.Box = Discrete, Box
    gym.Env, gym.spaces = Env, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    nc = types.ModuleType("netCDF4")
    nc.Dataset = object
    sys.modules["netCDF4"] = nc


def load(version):
    """version in {'PPOV1.1','PPOV2.0','PPOV2.1'} -> (config, environment, model, train)."""
    sys.dont_write_bytecode = True
    _install_third_party_stubs()
    for m in _MODS:
        sys.modules.pop(m, None)
    d = f"{REF_ROOT}/{version}"
    sys.path.insert(0, d)
    try:
        import config
        import environment
        import model
        script = "train_ppo1.1.py" if version == "PPOV1.1" else "train_ppo2.0.py"
        spec = importlib.util.spec_from_file_location("train_ref", f"{d}/{script}")
        train = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(train)
    finally:
        sys.path.remove(d)
    mods = (config, environment, model, train)
    for m in _MODS:
        sys.modules.pop(m, None)
    return mods
