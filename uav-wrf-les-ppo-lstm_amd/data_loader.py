"""data_loader.py -- counterpart of the reference's PPOV2.0/data_loader.py:5-22: concentration sequences and source
concentrations of the logged episodes.  Reads the reference's netCDF file when netCDF4 is installed, or an .npz with the
same variable names (x, concentration, source_concentration; NaN = unused step) otherwise."""
import numpy as np


def _arrays(path):
    if str(path).endswith(".npz"):
        d = np.load(path)
        return d["x"], d["concentration"], d["source_concentration"]
    from netCDF4 import Dataset          # same calls as the reference
    with Dataset(path, "r") as nc:
        return (np.ma.filled(nc["x"][:], np.nan), np.ma.filled(nc["concentration"][:], np.nan),
                np.ma.filled(nc["source_concentration"][:], np.nan))


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
