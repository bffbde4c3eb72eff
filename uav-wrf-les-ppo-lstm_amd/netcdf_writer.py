"""netcdf_writer.py -- counterpart of the reference's PPOV2.0/netcdf_writer.py:4-114: the episode x step trajectory log that
hands successful episodes from PPO training to the LSTM stage (loaded back by data_loader.load_raw_sequences).

Same constructor, variables (episode, step, x, y, concentration, is_source, source_concentration, source_x, source_y),
dtypes, fill values and write_episode_data() semantics (the last step's x / y are overwritten with the source coordinates
and flagged in is_source).  Three back ends, chosen by what is installed and by the file name:
  * netCDF4 installed, name not ending in .npz: a real NETCDF4 file written through the reference's own calls;
  * no netCDF4 (this image), name not ending in .npz: the arrays are kept in memory and close() writes a real netCDF file in the
    CLASSIC format (NetCDF-3, 64-bit offsets) through scipy.io.netcdf_file -- same dimensions, variable names, dtypes,
    `_FillValue` / `long_name` / `units` attributes and GRID_SIZE global attribute; the netCDF library (hence the reference's
    `netCDF4.Dataset(path)` in data_loader.py:5-22 and PPOV2.1/model.py:68-90) reads classic files transparently, only zlib
    compression and the HDF5 container are missing;
  * name ending in .npz: an .npz with the same variable names (tests, quick runs).  PPOV2.1 keeps the same class in model.py:355-423 with two more per-episode variables,
gaussian_sigma and peak_concentration, and two more write_episode_data() arguments: both are here, optional."""
from __future__ import annotations

import json

import numpy as np

# variable attributes of netcdf_writer.py:31-85 / PPOV2.1/model.py:369-398 (the schema PPOV2.1/nc_info.txt:1-46 shows)
ATTRS = {
    "episode": {"long_name": "Training episode index"},
    "step": {"long_name": "Step index within episode"},
    "x": {"units": "grid unit", "long_name": "Agent x-coordinate"},
    "y": {"units": "grid unit", "long_name": "Agent y-coordinate"},
    "concentration": {"long_name": "Methane concentration"},
    "is_source": {"long_name": "Source position flag"},
    "source_concentration": {"long_name": "Actual source concentration in each episode"},
    "source_x": {"long_name": "Actual source x-coordinate"},
    "source_y": {"long_name": "Actual source y-coordinate"},
    "gaussian_sigma": {"long_name": "Gaussian distribution standard deviation"},
    "peak_concentration": {"units": "ppm", "long_name": "Source peak concentration"},
}
# netCDF's default fill values, what a variable created WITHOUT fill_value holds where nothing was written
# (episode, step, gaussian_sigma, peak_concentration in the reference's writer)
NC_FILL_FLOAT, NC_FILL_INT = np.float32(9.969209968386869e36), np.int32(-2147483647)


class NetCDFWriter:
    def __init__(self, filename, grid_size, max_episodes=2000, max_steps=1000):
        self.filename, self.max_episodes, self.max_steps, self.grid_size = filename, max_episodes, max_steps, grid_size
        self._nc = None
        if not str(filename).endswith(".npz"):
            try:
                from netCDF4 import Dataset
                self._nc = Dataset(filename, mode="w", format="NETCDF4")
            except ImportError:
                pass                    # classic-format file through scipy at close()
        E, S = max_episodes, max_steps
        if self._nc is not None:
            nc = self._nc
            nc.createDimension("episode", E)
            nc.createDimension("step", S)
            nc.GRID_SIZE = grid_size
            self.episode_var = nc.createVariable("episode", np.int32, ("episode",))
            self.step_var = nc.createVariable("step", np.int32, ("step",))
            mk = lambda name, dt, dims, fill: nc.createVariable(name, dt, dims, fill_value=fill, zlib=True)
            self.x_var = mk("x", np.float32, ("episode", "step"), np.nan)
            self.y_var = mk("y", np.float32, ("episode", "step"), np.nan)
            self.conc_var = mk("concentration", np.float32, ("episode", "step"), np.nan)
            self.source_var = mk("is_source", np.int8, ("episode", "step"), 0)
            self.source_conc_var = mk("source_concentration", np.float32, ("episode",), np.nan)
            self.source_x_var = mk("source_x", np.float32, ("episode",), np.nan)
            self.source_y_var = mk("source_y", np.float32, ("episode",), np.nan)
            self.sigma_var = nc.createVariable("gaussian_sigma", np.float32, ("episode",))
            self.peak_var = nc.createVariable("peak_concentration", np.float32, ("episode",))
            for name, attrs in ATTRS.items():
                for k, v in attrs.items():
                    setattr(nc.variables[name], k, v)
        else:
            self.episode_var = np.full(E, NC_FILL_INT, np.int32)
            self.step_var = np.full(S, NC_FILL_INT, np.int32)
            self.x_var = np.full((E, S), np.nan, np.float32)
            self.y_var = np.full((E, S), np.nan, np.float32)
            self.conc_var = np.full((E, S), np.nan, np.float32)
            self.source_var = np.zeros((E, S), np.int8)
            self.source_conc_var = np.full(E, np.nan, np.float32)
            self.source_x_var = np.full(E, np.nan, np.float32)
            self.source_y_var = np.full(E, np.nan, np.float32)
            self.sigma_var = np.full(E, NC_FILL_FLOAT, np.float32)
            self.peak_var = np.full(E, NC_FILL_FLOAT, np.float32)

    def write_episode_data(self, episode_idx, steps, x, y, conc, source_x, source_y, source_conc, sigma=None, peak=None):
        """netcdf_writer.py:87-110 (called for successful episodes only); sigma / peak: PPOV2.1/model.py:405-419."""
        self.x_var[episode_idx, :steps] = x
        self.y_var[episode_idx, :steps] = y
        self.conc_var[episode_idx, :steps] = conc
        self.source_var[episode_idx, steps - 1] = 1
        self.x_var[episode_idx, steps - 1] = source_x
        self.y_var[episode_idx, steps - 1] = source_y
        self.source_conc_var[episode_idx] = source_conc
        self.source_x_var[episode_idx] = source_x
        self.source_y_var[episode_idx] = source_y
        if sigma is not None:
            self.sigma_var[episode_idx] = sigma
        if peak is not None:
            self.peak_var[episode_idx] = peak

    def _write_classic(self):
        """NetCDF-3 classic (64-bit offset) image of the arrays: what netCDF4.Dataset(path, "r") reads back as the same
        variables (masked where the value equals `_FillValue`, as with the NETCDF4 file)."""
        from scipy.io import netcdf_file
        fills = {"x": np.float32(np.nan), "y": np.float32(np.nan), "concentration": np.float32(np.nan), "is_source": np.int8(0),
                 "source_concentration": np.float32(np.nan), "source_x": np.float32(np.nan), "source_y": np.float32(np.nan)}
        data = {"episode": (self.episode_var, ("episode",)), "step": (self.step_var, ("step",)),
                "x": (self.x_var, ("episode", "step")), "y": (self.y_var, ("episode", "step")),
                "concentration": (self.conc_var, ("episode", "step")), "is_source": (self.source_var, ("episode", "step")),
                "source_concentration": (self.source_conc_var, ("episode",)), "source_x": (self.source_x_var, ("episode",)),
                "source_y": (self.source_y_var, ("episode",)), "gaussian_sigma": (self.sigma_var, ("episode",)),
                "peak_concentration": (self.peak_var, ("episode",))}
        with netcdf_file(self.filename, "w", version=2) as nc:
            nc.createDimension("episode", self.max_episodes)
            nc.createDimension("step", self.max_steps)
            nc.GRID_SIZE = np.int32(self.grid_size)
            for name, (arr, dims) in data.items():
                v = nc.createVariable(name, arr.dtype.char if arr.dtype != np.int8 else "b", dims)
                if name in fills:
                    v._FillValue = fills[name]
                for k, val in ATTRS[name].items():
                    setattr(v, k, val)
                v[:] = arr

    def close(self):
        if self._nc is not None:
            self._nc.close()
            return
        if not str(self.filename).endswith(".npz"):
            self._write_classic()
            return
        np.savez_compressed(self.filename, episode=self.episode_var, step=self.step_var, x=self.x_var, y=self.y_var,
                            concentration=self.conc_var, is_source=self.source_var,
                            source_concentration=self.source_conc_var, source_x=self.source_x_var,
                            source_y=self.source_y_var, gaussian_sigma=self.sigma_var, peak_concentration=self.peak_var,
                            GRID_SIZE=np.int64(self.grid_size), attrs_json=np.asarray(json.dumps(ATTRS)))
