"""data_loader.py -- counterpart of the reference's PPOV2.0/data_loader.py:5-22 (concentration sequences and source
concentrations of the logged episodes) and of load_trajectory_segments, PPOV2.1/model.py:68-90 (sliding windows).  Reads the reference's netCDF file when netCDF4 is installed, or an .npz with the
same variable names (x, concentration, source_concentration; NaN = unused step); without netCDF4 a CLASSIC-format netCDF
file (what netcdf_writer.py writes in that case, or any NetCDF-3 file with these variables) is read through scipy.io."""
import numpy as np


def _nc_variables(path, names):
    """{name: float/int array with fills as NaN} for the variables of `names` the file holds."""
    try:
        from netCDF4 import Dataset          # same calls as the reference
    except ImportError:
        from scipy.io import netcdf_file     # classic format only (an HDF5-based NETCDF4 file needs the netCDF4 package)
        with netcdf_file(path, "r", mmap=False) as nc:
            out = {}
            for k in names:
                if k in nc.variables:
                    v = nc.variables[k]
                    a = np.array(v[:])
                    fill = getattr(v, "_FillValue", None)
                    if fill is not None and a.dtype.kind == "f" and not np.isnan(fill):
                        a = np.where(a == fill, np.nan, a)
                    out[k] = a
            return out
    with Dataset(path, "r") as nc:
        return {k: np.ma.filled(nc[k][:], np.nan) for k in names if k in nc.variables}


def _arrays(path):
    if str(path).endswith(".npz"):
        d = np.load(path)
        return d["x"], d["concentration"], d["source_concentration"]
    v = _nc_variables(path, ("x", "concentration", "source_concentration"))
    return v["x"], v["concentration"], v["source_concentration"]


def load_raw_sequences(nc_path):
    x, conc, src = _arrays(nc_path)
    sequences, source_concs = [], []
    for ep in range(x.shape[0]):
        steps = np.where(~np.isnan(x[ep]))[0]
        if len(steps) == 0:
            continue
        sequences.append(conc[ep, :steps[-1] + 1].tolist())
        source_concs.append(src[ep])
    return sequences, np.array(source_concs)


def load_trajectory_segments(nc_path, tail_steps=60, window_size=20):
    """PPOV2.1/model.py:68-90: every length-`window_size` sliding window of every episode with at least that many logged
    steps (tail_steps is accepted and unused, as in the reference)."""
    if str(nc_path).endswith(".npz"):
        d = np.load(nc_path)
        get = lambda k: d[k] if k in d.files else None
    else:
        held = _nc_variables(nc_path, ("x", "y", "concentration", "source_x", "source_y", "gaussian_sigma"))
        get = held.get
    x, y, conc, sx, sy, sig = (get(k) for k in ("x", "y", "concentration", "source_x", "source_y", "gaussian_sigma"))
    segments = []
    for ep in range(x.shape[0]):
        steps = np.where(~np.isnan(x[ep]))[0]
        if len(steps) < window_size:
            continue
        xs, ys, cs = x[ep, steps], y[ep, steps], conc[ep, steps]
        source_pos = np.array([sx[ep], sy[ep]])
        sigma = sig[ep] if sig is not None else 15.0
        for i in range(0, len(steps) - window_size + 1):
            segments.append({"positions": np.column_stack((xs[i:i + window_size], ys[i:i + window_size])),
                             "concentrations": cs[i:i + window_size], "source_pos": source_pos, "sigma": sigma})
    print(f"Generated {len(segments)} segments (window_size={window_size})")
    return segments
