"""N4: trajectory-log oracle.  RadiusTracker is pinned to the reference's class (golden trace); the NetCDF writer /
loader are restated from source (netCDF4 is absent here: parity unpinned) and checked for their stated semantics.  CPU only."""
import os
import sys

import numpy as np

from oracle import traj_oracle as to

GOLD = os.path.join(os.path.dirname(__file__), "golden", "curriculum.npz")
ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uav-wrf-les-ppo-lstm_amd")


def test_radius_tracker_matches_reference_trace():
    g = np.load(GOLD, allow_pickle=False)
    rt = to.RadiusTrackerOracle()
    for k, (r, s) in enumerate(zip(g["tracker_radii"], g["tracker_success"])):
        rt.update(float(r), {"r": float(r)}, bool(s))
        want = g["tracker_history"][k]
        got = (rt.radius_history + [np.nan, np.nan])[:2]
        assert np.allclose(got, want, equal_nan=True), (k, got, want)
    counts = np.asarray([[k, len(v)] for k, v in sorted(rt.success_data.items())], np.float64)
    assert np.array_equal(counts, g["tracker_counts"])


def test_writer_semantics_and_loader_round_trip():
    a = to.writer_arrays(6, 12)
    rng = np.random.RandomState(0)
    eps = {}
    for ep, steps in ((0, 5), (2, 12), (3, 1)):
        x, y, c = rng.rand(steps) * 499, rng.rand(steps) * 499, rng.rand(steps) * 100
        to.write_episode(a, ep, steps, x, y, c, 123.5, 321.25, 77.0)
        eps[ep] = (steps, x, y, c)
    for ep, (steps, x, y, c) in eps.items():
        assert a["is_source"][ep].sum() == 1 and a["is_source"][ep, steps - 1] == 1
        assert a["x"][ep, steps - 1] == np.float32(123.5) and a["y"][ep, steps - 1] == np.float32(321.25)   # overwritten by the source
        assert np.allclose(a["x"][ep, :steps - 1], x[:steps - 1].astype(np.float32)) and np.isnan(a["x"][ep, steps:]).all()
        assert np.allclose(a["concentration"][ep, :steps], c.astype(np.float32))
    seqs, concs = to.load_raw_sequences(a)
    assert [len(s) for s in seqs] == [5, 12, 1] and np.allclose(concs, 77.0)       # unwritten episodes are skipped
    assert np.allclose(seqs[1], eps[2][3].astype(np.float32))


def test_product_writer_and_loader_agree_with_the_oracle(tmp_path):
    """The product's NetCDFWriter (npz back end here) and data_loader follow the same semantics."""
    sys.path.insert(0, ROOT)
    try:
        from data_loader import load_raw_sequences
        from netcdf_writer import NetCDFWriter
    finally:
        sys.path.remove(ROOT)
    path = str(tmp_path / "training_data.npz")
    w = NetCDFWriter(path, 500, max_episodes=5, max_steps=9)
    a = to.writer_arrays(5, 9)
    rng = np.random.RandomState(1)
    for ep, steps in ((1, 4), (4, 9)):
        x, y, c = rng.rand(steps) * 499, rng.rand(steps) * 499, rng.rand(steps) * 100
        args = (ep, steps, x, y, c, float(x[-1]), float(y[-1]), float(c[-1]))
        w.write_episode_data(*args)
        to.write_episode(a, *args)
    w.close()
    d = np.load(path)
    for k in a:
        assert np.array_equal(d[k], a[k], equal_nan=True), k
    assert int(d["GRID_SIZE"]) == 500
    seqs, concs = load_raw_sequences(path)
    oseqs, oconcs = to.load_raw_sequences(a)
    assert len(seqs) == 2 and all(np.allclose(s, o) for s, o in zip(seqs, oseqs)) and np.allclose(concs, oconcs)


def test_v21_writer_fields_and_segment_loader(tmp_path):
    """PPOV2.1's extra per-episode variables and load_trajectory_segments (model.py:68-90): product (npz back end) == oracle."""
    sys.path.insert(0, ROOT)
    try:
        from data_loader import load_trajectory_segments
        from netcdf_writer import NetCDFWriter
    finally:
        sys.path.remove(ROOT)
    path = str(tmp_path / "training_data.npz")
    w = NetCDFWriter(path, 500, max_episodes=4, max_steps=40)
    a = to.writer_arrays(4, 40)
    rng = np.random.RandomState(2)
    for ep, steps in ((0, 19), (1, 20), (3, 27)):
        x, y, c = rng.rand(steps) * 499, rng.rand(steps) * 499, rng.rand(steps) * 100
        args = (ep, steps, x, y, c, 40.0 + ep, 50.0 + ep, 100.0)
        w.write_episode_data(*args, sigma=15.0, peak=100.0)
        to.write_episode(a, *args, sigma=15.0, peak=100.0)
    w.close()
    d = np.load(path)
    for k in a:
        assert np.array_equal(d[k], a[k], equal_nan=True), k
    segs, osegs = load_trajectory_segments(path, tail_steps=60), to.load_trajectory_segments(a)
    assert len(segs) == len(osegs) == 1 + 8                                   # 19 steps: none; 20: one; 27: eight windows
    for s_, o in zip(segs, osegs):
        assert np.array_equal(s_["positions"], o["positions"]) and np.array_equal(s_["concentrations"], o["concentrations"])
        assert np.array_equal(s_["source_pos"], o["source_pos"]) and s_["sigma"] == o["sigma"] == np.float32(15.0)
    assert np.array_equal(segs[-1]["positions"][-1], [43.0, 53.0])              # the last logged step carries the source position
