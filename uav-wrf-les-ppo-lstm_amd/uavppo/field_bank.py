"""Materialised plume-field banks for the vectorised environments (BASELINE C4: "WRF-LES training_data.nc wind field").

A bank is `[F, 500, 500, 2]` f64 (concentration, tke) resident in HBM plus the `[F, 2]` source position of every field;
episode k of global env e uses field (e + k * N_total) mod F (csrc/env_core.h).  Three ways to get one:

* a file: `.npz` with arrays `conc` [F, 500, 500], `tke` [F, 500, 500], `source` [F, 2] (`conc` on the reference's 0..100
  scale, the tables its `_generate_plume` builds, environment.py:51-62), or a netCDF file with variables of the same names
  when the netCDF4 package is installed (it is not in this image; the reference names no schema for a field file -- its
  `training_data.nc` is a TRAJECTORY log, PPOV2.1/nc_info.txt -- so the names are this build's);
* `"synth:F"`: F fields generated on the device by the procedural sampler itself with E3's formula (`uav_env_materialise`),
  the generator bench.py's C4 configuration uses;
* arrays handed over directly.
"""
from __future__ import annotations

import numpy as np
import torch

GRID = 500


def _check(conc, tke, source):
    conc, tke, source = (np.asarray(a, dtype=np.float64) for a in (conc, tke, source))
    if conc.ndim != 3 or conc.shape[1:] != (GRID, GRID) or tke.shape != conc.shape:
        raise ValueError(f"field bank: conc / tke must be [F, {GRID}, {GRID}], got {conc.shape} / {tke.shape}")
    if source.shape != (conc.shape[0], 2):
        raise ValueError(f"field bank: source must be [F, 2], got {source.shape}")
    if not (np.isfinite(conc).all() and np.isfinite(tke).all() and np.isfinite(source).all()):
        raise ValueError("field bank: non-finite values")
    return conc, tke, source


def from_arrays(conc, tke, source, device="cuda"):
    conc, tke, source = _check(conc, tke, source)
    bank = torch.from_numpy(np.stack([conc, tke], axis=-1)).to(device).contiguous()
    return bank, torch.from_numpy(source).to(device).contiguous()


def synthesise(n_fields, variant="v2.1", device="cuda", seed=4321):
    """F fields by E3's formula from the counter RNG, entirely on the device (no host tables)."""
    from . import ops
    from .vec_env import VecMethaneEnv
    gen = VecMethaneEnv(int(n_fields), variant, device, seed=seed)
    gen.reset()
    bank = torch.stack([ops.env_materialise(gen.state, int(n_fields), gen.cfg(), f) for f in range(int(n_fields))])
    return bank, gen.peek()[1]


def save_npz(path, bank, sources):
    b = torch.as_tensor(bank).cpu().numpy()
    np.savez_compressed(path, conc=b[..., 0], tke=b[..., 1], source=torch.as_tensor(sources).cpu().numpy())


def load(spec, variant="v2.1", device="cuda"):
    """spec: None | "synth:F" | path (.npz, or netCDF when the package exists) | (conc, tke, source) arrays  ->  (bank, sources) or (None, None)."""
    if spec is None or spec == "":
        return None, None
    if isinstance(spec, (tuple, list)):
        return from_arrays(*spec, device=device)
    spec = str(spec)
    if spec.startswith("synth:"):
        return synthesise(int(spec.split(":", 1)[1]), variant, device)
    if spec.endswith(".npz"):
        with np.load(spec) as d:
            return from_arrays(d["conc"], d["tke"], d["source"], device=device)
    try:
        from netCDF4 import Dataset
    except ImportError as e:
        raise RuntimeError(f"field bank {spec!r}: netCDF4 is not installed here; convert the file to .npz (conc, tke, source)") from e
    with Dataset(spec, "r") as nc:
        return from_arrays(np.ma.filled(nc["conc"][:], np.nan), np.ma.filled(nc["tke"][:], np.nan),
                           np.ma.filled(nc["source"][:], np.nan), device=device)
