"""train_ppo2.0.py writes its per-episode CSV through the csv module: the file must be the one the reference's
pd.DataFrame(...).to_csv(path, index=False) writes (train_ppo2.0.py:257-258), byte for byte."""
import importlib.util
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")


def _script():
    sys.path[:0] = [p for p in (ROOT, PKG) if p not in sys.path]
    spec = importlib.util.spec_from_file_location("train_ppo2_0_csv", os.path.join(PKG, "train_ppo2.0.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rows(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, 11)) * rng.choice([1e-9, 1e-3, 1.0, 1e4, 1e17], size=(n, 1))
    rows = [[k + 1, r[1], int(abs(r[2])) % 2, *r[3:8], int(abs(r[8])) % 300, *r[9:]] for k, r in enumerate(a.tolist())]
    rows[0][1], rows[1][1], rows[2][1], rows[3][1], rows[4][1] = 0.1 + 0.2, -0.0, 3.0, 1.5e22, 5e-324
    return rows


def test_csv_writer_equals_the_dataframe_writer(tmp_path):
    m = _script()
    for case, rows in (("plain", _rows(3000, 0)), ("empty", [])):
        a, b = tmp_path / f"{case}_a.csv", tmp_path / f"{case}_b.csv"
        m._write_csv(rows, str(a))
        pd.DataFrame(rows, columns=m.COLUMNS).to_csv(str(b), index=False)
        assert a.read_bytes() == b.read_bytes(), case


def test_csv_writer_leaves_odd_values_to_pandas(tmp_path):
    m = _script()
    for case, v in (("nan", float("nan")), ("inf", float("inf")), ("numpy scalar", np.float64(2.5)), ("numpy int", np.int64(7))):
        rows = _rows(50, 1)
        rows[7][4] = v
        a, b = tmp_path / "a.csv", tmp_path / "b.csv"
        m._write_csv(rows, str(a))
        pd.DataFrame(rows, columns=m.COLUMNS).to_csv(str(b), index=False)
        assert a.read_bytes() == b.read_bytes(), case


def test_streamed_csv_equals_the_dataframe_writer(tmp_path):
    m = _script()
    rows = _rows(700, 2)
    for case, limit, poison in (("all", None, None), ("cut", 450, None), ("nan late", None, 650), ("numpy late", 450, 300)):
        rs = [list(r) for r in rows]
        if poison is not None:
            rs[poison][5] = float("nan") if "nan" in case else np.float64(1.25)
        a, b = tmp_path / "a.csv", tmp_path / "b.csv"
        st = m._CsvStream(str(a), limit)
        for i in range(0, len(rs), 128):
            st.add(rs[i:i + 128])
        kept = rs[:limit] if limit is not None else rs
        st.close(kept)
        pd.DataFrame(kept, columns=m.COLUMNS).to_csv(str(b), index=False)
        assert a.read_bytes() == b.read_bytes(), case
