"""Import the reference's hot-path modules in the BUILD container (never on the GPU box).

Only `oracle/gen_golden.py` uses this.  `gym` and `netCDF4` are absent from the image and
contribute no arithmetic (base class + space objects / an output writer), so they are
replaced by empty stand-ins; the reference's own code runs untouched from /root/reference.
"""
import importlib.util
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"

# ---- in-memory stand-in for netCDF4.Dataset (the package is absent from the image).  It holds what the reference's writer
# and loaders DO with a Dataset -- dimensions, variables with dtype / fill value / attributes, slice assignment, reads -- in
# numpy arrays, so that the reference's own NetCDFWriter / load_raw_sequences / load_trajectory_segments code can be run and
# its results recorded as golden vectors (oracle/gen_golden.py traj).  netCDF's on-disk encoding is not modelled.
NC_DEFAULT_FILL = {np.dtype(np.float32): np.float32(9.969209968386869e36), np.dtype(np.int32): np.int32(-2147483647),
                   np.dtype(np.int8): np.int8(-127)}
_MEM_FILES = {}


class MemVariable:
    def __init__(self, name, dtype, dims, shape, fill_value):
        object.__setattr__(self, "_attrs", {})
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "dimensions", tuple(dims))
        object.__setattr__(self, "explicit_fill", fill_value is not None)
        dt = np.dtype(dtype)
        fill = NC_DEFAULT_FILL[dt] if fill_value is None else dt.type(fill_value)
        object.__setattr__(self, "fill_value", fill)
        object.__setattr__(self, "data", np.full(shape, fill, dt))

    def __setattr__(self, k, v):            # var.long_name = "..." -> a netCDF attribute
        self._attrs[k] = v

    def __setitem__(self, idx, val):
        self.data[idx] = val

    def __getitem__(self, idx):
        return self.data[idx]

    def __len__(self):
        return len(self.data)


class MemDataset:
    def __init__(self, filename, mode="r", format=None):
        if mode == "r":
            src = _MEM_FILES[filename]
            self.__dict__.update(src.__dict__)
            return
        object.__setattr__(self, "dimensions", {})
        object.__setattr__(self, "variables", {})
        object.__setattr__(self, "_gattrs", {})
        _MEM_FILES[filename] = self

    def __setattr__(self, k, v):            # ncfile.GRID_SIZE = ... -> a global attribute
        self._gattrs[k] = v

    def createDimension(self, name, size):
        self.dimensions[name] = size

    def createVariable(self, name, dtype, dims, fill_value=None, zlib=False):
        v = MemVariable(name, dtype, dims, tuple(self.dimensions[d] for d in dims), fill_value)
        self.variables[name] = v
        return v

    def __getitem__(self, name):
        return self.variables[name]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

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
    nc.Dataset = MemDataset
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
