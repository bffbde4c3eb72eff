"""N3: the stop-predictor training oracle vs goldens recorded from the reference's own SequenceDataset, model,
SmoothL1Loss, AdamW and clipping (oracle/gen_golden.py train_lstm), and vs torch's ReduceLROnPlateau.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import lstm_train_oracle as lt

GOLD = os.path.join(os.path.dirname(__file__), "golden", "train_lstm_v20.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def _seqs(g):
    out, o = [], 0
    for L in g["seq_lens"]:
        out.append(list(g["seq_flat"][o:o + L]))
        o += L
    return out


def test_sequence_dataset_matches_reference(gold):
    X, Y, lo, hi = lt.sequence_dataset(_seqs(gold), gold["source_concs"], int(gold["training_size"]))
    assert X.shape == gold["X"].shape and np.allclose(X, gold["X"], atol=1e-6) and np.allclose(Y, gold["Y"], atol=1e-6)
    assert np.isclose(lo, gold["data_min"][0]) and np.isclose(hi, gold["data_max"][0])
    assert (gold["seq_lens"] < int(gold["training_size"])).any()          # short sequences are dropped, as in the reference


def test_three_optimiser_steps_match_reference(gold):
    params = {k[5:]: torch.from_numpy(gold[k].copy()) for k in gold.files if k.startswith("init/")}
    opt = lt.AdamWState(params, lr=3e-4)
    x, y = torch.from_numpy(gold["X"][:24])[:, :, None], torch.from_numpy(gold["Y"][:24])
    for k in range(3):
        loss, gn = lt.train_step(params, opt, x, y, masks=None)
        assert np.isclose(loss, gold["losses"][k], rtol=2e-6), (k, loss, gold["losses"][k])
        assert np.isclose(gn, gold["gnorms"][k], rtol=2e-5)
    for k, v in params.items():
        want = gold["post/" + k]
        assert np.allclose(v.numpy(), want, rtol=1e-4, atol=2e-7), k


def test_three_optimiser_steps_on_ragged_sequences_match_reference(gold):
    """The reference's forward(x, lengths) -- pack_padded_sequence, the output at step lengths[i] - 1 of sequence i
    (PPOV2.0/model.py:229-240) -- for three optimiser steps on zero-padded sequences of lengths 1..10."""
    params = {k[5:]: torch.from_numpy(gold[k].copy()) for k in gold.files if k.startswith("init/")}
    opt = lt.AdamWState(params, lr=3e-4)
    x, y = torch.from_numpy(gold["ragged_x"])[:, :, None], torch.from_numpy(gold["Y"][:24])
    lens = gold["ragged_lengths"]
    assert lens.min() == 1 and lens.max() == x.shape[1] and len(set(lens.tolist())) >= 6
    for k in range(3):
        loss, gn = lt.train_step(params, opt, x, y, masks=None, lengths=lens)
        assert np.isclose(loss, gold["ragged_losses"][k], rtol=2e-6), (k, loss, gold["ragged_losses"][k])
        assert np.isclose(gn, gold["ragged_gnorms"][k], rtol=2e-5)
    for k, v in params.items():
        assert np.allclose(v.numpy(), gold["ragged_post/" + k], rtol=1e-4, atol=2e-7), k
    # what lies behind a sequence's end cannot matter
    x2 = x.clone()
    for i, L in enumerate(lens):
        x2[i, L:] = 7.0
    with torch.no_grad():
        assert torch.equal(lt.predictor_forward(params, x, None, lens), lt.predictor_forward(params, x2, None, lens))


def test_reduce_lr_on_plateau_matches_torch(gold):
    sch = lt.ReduceLROnPlateauOracle(3e-4)
    lrs = [sch.step(float(m)) for m in gold["sched_metrics"]]
    assert np.allclose(lrs, gold["sched_lrs"], rtol=0, atol=1e-15) and len(set(lrs)) >= 4


def test_dropout_masks_of_ones_are_eval_mode(gold):
    params = {k[5:]: torch.from_numpy(gold[k].copy()) for k in gold.files if k.startswith("init/")}
    x = torch.from_numpy(gold["X"][:5])[:, :, None]
    H = params["lstm.weight_hh_l0"].shape[1]
    ones = {"l0": torch.ones(5, x.shape[1], H), "l1": torch.ones(5, x.shape[1], H), "head": torch.ones(5, 64)}
    with torch.no_grad():
        assert torch.equal(lt.predictor_forward(params, x, None), lt.predictor_forward(params, x, ones))


# ---- V2.1 variant (PPOV2.1/train_lstm.py): TrajectoryDataset + MSE/BCE loop body
GOLD21 = os.path.join(os.path.dirname(__file__), "golden", "train_lstm_v21.npz")


def segments_of(g):
    return [{"positions": g["positions"][i], "concentrations": g["concentrations"][i], "source_pos": g["source_pos"][i]}
            for i in range(int(g["n_seg"]))]


def test_trajectory_dataset_matches_reference():
    import random
    g = np.load(GOLD21, allow_pickle=False)
    X, labels = lt.trajectory_dataset(segments_of(g), stop_radius=10, window_size=int(g["window"]), rng=random.Random(5))
    assert X.shape[0] == 60 and np.array_equal(X[:, :, 0], g["X"]) and np.array_equal(labels, g["labels"])
    assert 0 < labels[:, 1].sum() < 30 and (labels[0::2, 1] == 0).all()       # negatives first of each pair; both stop labels occur


def test_peak_stop_three_steps_match_reference():
    g = np.load(GOLD21, allow_pickle=False)
    params = {k[5:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("init/")}
    opt = lt.AdamWState(params, lr=1e-3, weight_decay=1e-4)
    x = torch.tensor(g["X"][:32], dtype=torch.float32)[:, :, None]
    y = torch.tensor(g["labels"][:32], dtype=torch.float32)
    for k in range(3):
        loss, gn = lt.peak_stop_train_step(params, opt, x, y)
        assert np.isclose(loss, g["losses"][k], rtol=2e-6) and np.isclose(gn, g["gnorms"][k], rtol=2e-5)
    for k, v in params.items():
        assert np.allclose(v.numpy(), g["post/" + k], rtol=1e-4, atol=2e-7), k
