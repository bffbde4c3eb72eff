"""The drop-in script (uav-wrf-les-ppo-lstm_amd/train_ppo2.0.py, counterpart of the reference's PPOV2.0/train_ppo2.0.py:110-261)
must be able to run every BASELINE configuration from config.py knobs alone: a materialised field bank loaded from a FILE
(C4), trend channels + stacked layers (C5), several ranks (env shards, gradient all-reduce, rank-0-only CSV / checkpoint).
Two gloo ranks sharing this box's GPU must reproduce the one-rank run.  -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import importlib.util, json, os, sys, torch
sys.path[:0] = [ROOT, PKG]
import config
over = json.loads(os.environ["CFG"])
world = int(os.environ.get("WORLD_SIZE", "1"))
over["NUM_ENVS"] = over["NUM_ENVS"] // world            # the job's envs split over the ranks
for k, v in over.items():
    assert hasattr(config, k), k
    setattr(config, k, v)
spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(PKG, "train_ppo2.0.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
out = os.environ["OUT"]
eps = os.environ.get("EPISODES")
if eps:       # stop at an episode count (every rank must leave the loop after the same iteration)
    tr, rows = m.train_ppo_vectorised(episodes=int(eps), csv_path=out + ".csv", model_path=out + ".pth", log_every=3)
else:
    tr, rows = m.train_ppo_vectorised(iterations=int(os.environ.get("ITERS", "2")), csv_path=out + ".csv", model_path=out + ".pth", log_every=1)
torch.save({"flat": tr.policy.flat.cpu(), "rows": rows, "radius": tr.radius, "episodes": tr.episodes_done, "obs": tr.buf["obs"].cpu(),
            "info": tr.info.cpu(), "iteration": tr.iteration}, out + f".{tr.rank}")
import torch.distributed as dist
if dist.is_initialized():
    dist.destroy_process_group()
'''


def _run(world, out, port, cfg, iters=2, **extra):
    import json
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OUT=out, CFG=json.dumps(cfg), ITERS=str(iters), UAVPPO_DIST_BACKEND="gloo", GPU_MAX_HW_QUEUES="2", **extra)
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT={ROOT!r}; PKG={PKG!r}\n" + WORKER], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0


@pytest.fixture(scope="module")
def bank_file(tmp_path_factory):
    """A field file in the build's schema (conc / tke / source), written from four synthesised PPOV2.1 fields."""
    from uavppo import field_bank
    bank, src = field_bank.synthesise(4, "v2.1", "cuda:0", seed=99)
    path = str(tmp_path_factory.mktemp("bank") / "fields.npz")
    field_bank.save_npz(path, bank, src)
    back, src2 = field_bank.load(path, "v2.1", "cuda:0")
    assert torch.equal(back, bank) and torch.equal(src2, src)
    return path


CASES = {
    # BASELINE C4's family: materialised bank from a file, sigma = 15, LSTM h=128 (fused persistent rollout)
    "c4": dict(NUM_ENVS=64, HORIZON=32, POLICY="lstm", HIDDEN=128, NUM_LAYERS=1, ENV_VARIANT="v2.1", EPOCHS=2),
    # BASELINE C5's family: h = 256 stacked x2, two trend channels (stepper rollout, per-step kernels, split GEMMs)
    "c5": dict(NUM_ENVS=64, HORIZON=12, POLICY="lstm", HIDDEN=256, NUM_LAYERS=2, ENV_VARIANT="v2.1", TREND_K=2, EPOCHS=2),
}


@pytest.mark.parametrize("case", ["c4", "c5"])
def test_train_script_runs_the_config_and_two_ranks_equal_one(tmp_path, bank_file, case):
    cfg = dict(CASES[case])
    if case == "c4":
        cfg["FIELD_BANK"] = bank_file
    port = 29800 + os.getpid() % 1000 + 5 * list(CASES).index(case)
    _run(1, str(tmp_path / "w1"), port, cfg)
    _run(2, str(tmp_path / "w2"), port + 1, cfg)
    one = torch.load(tmp_path / "w1.0", weights_only=False)
    a, b = torch.load(tmp_path / "w2.0", weights_only=False), torch.load(tmp_path / "w2.1", weights_only=False)
    assert one["obs"].shape[2] == 6 + cfg.get("TREND_K", 0)
    assert torch.equal(torch.cat([a["obs"], b["obs"]], 0), one["obs"])          # same global RNG keys, same fields
    assert torch.equal(torch.cat([a["info"], b["info"]], 0), one["info"])
    assert torch.equal(a["flat"], b["flat"])
    diff = (a["flat"] - one["flat"]).abs()
    assert diff.max().item() < 0.1 * 4 * 3e-5 and diff.mean().item() < 2e-7     # see test_gpu_multirank.py
    assert a["radius"] == b["radius"] == one["radius"] and a["episodes"] == b["episodes"] == one["episodes"]
    # rows: rank 0 of the two-rank job merged both ranks' episodes in (iteration, global env, time) order
    assert a["rows"] == one["rows"] and len(one["rows"]) == one["episodes"]
    import pandas as pd
    df1, df2 = pd.read_csv(tmp_path / "w1.csv"), pd.read_csv(tmp_path / "w2.csv")
    assert list(df1.columns) == ["Episode", "Total_Reward", "Success", "Conc_Reward", "Explore_Reward", "Move_Penalty", "TKE_Penalty",
                                 "Boundary_Penalty", "Steps", "Final_Conc", "Current_Radius"]
    assert df1.equals(df2)
    sd = torch.load(tmp_path / "w2.pth")
    assert f"lstm.weight_hh_l{cfg['NUM_LAYERS'] - 1}" in sd
    if case == "c4":
        # the bank really is what the envs sample: obs[2] * 100 of a step is the file's conc at the agent's cell
        with np.load(bank_file) as d:
            conc, src = d["conc"], d["source"]
        info = one["info"].numpy()
        # columns 6..9: agent x, y and the episode's source -> which field; obs[2] in column 5
        hit = 0
        for n in range(0, 64, 7):
            f = int(np.argmin(np.abs(src - info[n, 0, 8:10]).sum(1)))
            assert np.allclose(src[f], info[n, 0, 8:10])
            x, y = int(info[n, 0, 6]), int(info[n, 0, 7])
            assert np.float32(conc[f, min(x, 499), min(y, 499)] / 100.0) == info[n, 0, 5]
            hit += 1
        assert hit >= 9


def test_train_script_stops_at_an_episode_count(tmp_path):
    """`episodes=` (the reference trains 2000, train_ppo2.0.py:128): the run ends once that many episodes finished and the
    CSV holds exactly that many rows, numbered from 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(PKG, "train_ppo2.0.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.NUM_ENVS, m.HORIZON, m.POLICY = 256, 64, "mlp"
    tr, rows = m.train_ppo_vectorised(episodes=40, csv_path=str(tmp_path / "e.csv"), model_path=None, log_every=0)
    assert len(rows) == 40 and [r[0] for r in rows] == list(range(1, 41)) and tr.episodes_done >= 40
    assert all(1 <= r[8] <= 1000 for r in rows)


def test_two_ranks_stop_together_at_an_episode_count(tmp_path):
    """`episodes=` with WORLD_SIZE = 2: the stop decision is taken from replicated device state (a fixed curriculum-mirror slot,
    trainer.episodes_before_rollout), not from each process's polled mirror -- so both ranks leave the loop after the SAME
    iteration (a rank stopping alone would strand the other in the next gradient all-reduce: this test would time out) and rank 0
    writes exactly `episodes` rows."""
    cfg = dict(NUM_ENVS=128, HORIZON=64, POLICY="lstm", HIDDEN=64, NUM_LAYERS=1, ENV_VARIANT="v2.0", EPOCHS=1)
    port = 29900 + os.getpid() % 1000
    _run(2, str(tmp_path / "e2"), port, cfg, EPISODES="60")
    a, b = torch.load(tmp_path / "e2.0", weights_only=False), torch.load(tmp_path / "e2.1", weights_only=False)
    assert a["iteration"] == b["iteration"] and a["iteration"] >= 3
    assert torch.equal(a["flat"], b["flat"]) and a["episodes"] == b["episodes"] >= 60
    assert len(a["rows"]) == 60 and [r[0] for r in a["rows"]] == list(range(1, 61))
