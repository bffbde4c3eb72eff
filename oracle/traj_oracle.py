"""TEST INFRASTRUCTURE -- CPU restatement of the reference's trajectory log.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this.

  writer_arrays / write_episode   NetCDFWriter.__init__/_init_variables/write_episode_data, PPOV2.0/netcdf_writer.py:4-110,
                                  as plain arrays (NaN / 0 fills; the last step's x, y replaced by the source coordinates)
  load_raw_sequences              PPOV2.0/data_loader.py:5-22 on those arrays
  RadiusTrackerOracle             train_ppo2.0.py:90-108
  (sigma=, peak=)                 PPOV2.1's writer (model.py:355-423): gaussian_sigma / peak_concentration per episode
  load_trajectory_segments        PPOV2.1/model.py:68-90 on those arrays (every sliding window of every long-enough episode)

Pinning: RadiusTracker against the reference's own class (tests/golden/curriculum.npz gets a `tracker_*` trace from
oracle/gen_golden.py curriculum).  Writer and loaders: tests/golden/traj_log.npz holds what the reference's OWN NetCDFWriter /
load_raw_sequences / load_trajectory_segments produce when run over an in-memory stand-in for netCDF4.Dataset
(oracle/_refload.MemDataset; the package is absent from the image), and tests/golden/nc_schema.json the reference-held schema
dump PPOV2.1/nc_info.txt; tests/test_oracle_traj.py checks this restatement and the product's writer against both.  Only
netCDF's on-disk encoding stays unpinned.
"""
import numpy as np


def writer_arrays(max_episodes, max_steps):
    E, S = max_episodes, max_steps
    return {"x": np.full((E, S), np.nan, np.float32), "y": np.full((E, S), np.nan, np.float32),
            "concentration": np.full((E, S), np.nan, np.float32), "is_source": np.zeros((E, S), np.int8),
            "source_concentration": np.full(E, np.nan, np.float32), "source_x": np.full(E, np.nan, np.float32),
            "source_y": np.full(E, np.nan, np.float32), "gaussian_sigma": np.full(E, np.float32(9.969209968386869e36), np.float32),
            "peak_concentration": np.full(E, np.float32(9.969209968386869e36), np.float32)}


def write_episode(a, episode_idx, steps, x, y, conc, source_x, source_y, source_conc, sigma=None, peak=None):
    a["x"][episode_idx, :steps] = x
    a["y"][episode_idx, :steps] = y
    a["concentration"][episode_idx, :steps] = conc
    a["is_source"][episode_idx, steps - 1] = 1
    a["x"][episode_idx, steps - 1] = source_x
    a["y"][episode_idx, steps - 1] = source_y
    a["source_concentration"][episode_idx] = source_conc
    a["source_x"][episode_idx] = source_x
    a["source_y"][episode_idx] = source_y
    if sigma is not None:
        a["gaussian_sigma"][episode_idx] = sigma
    if peak is not None:
        a["peak_concentration"][episode_idx] = peak


def load_trajectory_segments(a, window_size=20):
    segs = []
    for ep in range(a["x"].shape[0]):
        valid = np.where(~np.isnan(a["x"][ep]))[0]
        if len(valid) < window_size:
            continue
        xs, ys, cs = a["x"][ep, valid], a["y"][ep, valid], a["concentration"][ep, valid]
        src = np.array([a["source_x"][ep], a["source_y"][ep]])
        for i in range(0, len(valid) - window_size + 1):
            segs.append({"positions": np.column_stack((xs[i:i + window_size], ys[i:i + window_size])),
                         "concentrations": cs[i:i + window_size], "source_pos": src, "sigma": a["gaussian_sigma"][ep]})
    return segs


def load_raw_sequences(a):
    seqs, concs = [], []
    for ep in range(a["x"].shape[0]):
        steps = np.where(~np.isnan(a["x"][ep]))[0]
        if len(steps) == 0:
            continue
        seqs.append(a["concentration"][ep, :steps[-1] + 1].tolist())
        concs.append(a["source_concentration"][ep])
    return seqs, np.array(concs)


class RadiusTrackerOracle:
    def __init__(self):
        self.radius_history, self.success_data = [], {}

    def update(self, current_radius, episode_data, is_success):
        if is_success:
            self.success_data.setdefault(current_radius, []).append(episode_data)
            if current_radius not in self.radius_history:
                self.radius_history.append(current_radius)
                self.radius_history.sort()
                if len(self.radius_history) > 2:
                    del self.radius_history[-1]
