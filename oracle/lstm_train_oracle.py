"""TEST INFRASTRUCTURE -- CPU restatement of the reference's offline training of the stop predictor (SURVEY 8f row N3,
PPOV2.0/train_lstm.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

  sequence_dataset        SequenceDataset.__init__ / __getitem__, train_lstm.py:12-50 (last TRAINING_SIZE concentrations of every
                          long-enough sequence, one global MinMaxScaler over all windows, label = source concentration)
  predictor_forward       ConcentrationThresholdPredictor.forward in TRAIN mode with EXPLICIT dropout masks (nn.LSTM dropout
                          0.3 acts on the outputs of layers 0 and 1, nn.Dropout(0.1) after the head's ReLU; model.py:206-240);
                          masks of ones = eval mode
  train_step              zero_grad / forward / SmoothL1Loss(beta=2) / backward / clip_grad_norm_(1.0) / AdamW(lr, wd 0.01)
                          (train_lstm.py:66-67,86-92), gradients by torch autograd on the restated graph
  ReduceLROnPlateauOracle torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.5, patience=5), train_lstm.py:68-73

  trajectory_dataset      TrajectoryDataset._preprocess, PPOV2.1/train_lstm.py:28-66 (segments grouped by source position, up to
                          1000 groups drawn with random.sample, first segment of each: a negative sample = first window,
                          label (conc/100, 0), and a positive sample = last window, label (conc/100, within stop_radius))
  peak_stop_train_step    the V2.1 loop body, PPOV2.1/train_lstm.py:104-118: LSTM(1->H) h_n -> (peak, sigmoid stop),
                          MSELoss + BCELoss, clip_grad_norm_(1.0), AdamW(1e-3, wd 1e-4)

Pinning (tests/golden/train_lstm_v21.npz, gen_golden.py train_lstm_v21): the reference's TrajectoryDataset itself under
random.seed(5); three optimiser steps on the torch modules its (function-local) PeakAndStopPredictor is built from.
Pinning (tests/golden/train_lstm_v20.npz, oracle/gen_golden.py train_lstm): the reference's own SequenceDataset on synthetic
sequences, and three optimiser steps of its model / criterion / AdamW / clipping run in eval mode (dropout cannot be given
masks in the reference: with dropout on, the restatement is pinned by its structure only), on full-length windows AND on
ragged sequences through the reference's forward(x, lengths) (pack_padded_sequence, model.py:229-240).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import eval_oracle as eo
from . import ppo_oracle as po


def sequence_dataset(sequences, source_concs, training_size):
    windows = [np.asarray(s[-training_size:], np.float64) for s in sequences if len(s) >= training_size]
    labels = [float(c) for s, c in zip(sequences, source_concs) if len(s) >= training_size]
    if not windows:
        return np.zeros((0, training_size), np.float32), np.zeros(0, np.float32), np.nan, np.nan
    allv = np.concatenate(windows)
    lo, hi = float(allv.min()), float(allv.max())
    X = np.stack([eo.minmax_transform(np.asarray(w, np.float32), np.array([lo, hi])) for w in windows]).astype(np.float32)
    return X, np.asarray(labels, np.float32), lo, hi


def predictor_forward(p, x, masks=None, lengths=None):
    """p: dict of tensors with the reference's state_dict keys; x [B, T, 1]; masks: None (eval) or dict with
    'l0','l1' [B, T, H] and 'head' [B, 64], already scaled by 1/(1-p).  lengths (list / array of B ints, or None = all T):
    model.py:229-240 packs the sequences, so sequence i ends at step lengths[i] - 1 and THAT output feeds the head; an LSTM is
    causal, so running the padded steps too and picking the output at lengths[i] - 1 is the same function."""
    B = x.shape[0]
    seq = x.transpose(0, 1)
    for l in range(3):
        H = p[f"lstm.weight_hh_l{l}"].shape[1]
        z = torch.zeros(B, H, dtype=x.dtype)
        seq, hn, cn = po.lstm_layer_forward(seq, z, z, p[f"lstm.weight_ih_l{l}"], p[f"lstm.weight_hh_l{l}"],
                                            p[f"lstm.bias_ih_l{l}"], p[f"lstm.bias_hh_l{l}"], None)
        if masks is not None and l < 2:
            seq = seq * masks[f"l{l}"].transpose(0, 1)
    if lengths is None:
        h = seq[-1]
    else:
        h = seq[torch.as_tensor(np.asarray(lengths), dtype=torch.long) - 1, torch.arange(B)]
    z = F.linear(h, p["fc.0.weight"], p["fc.0.bias"])
    a = torch.relu(F.layer_norm(z, (z.shape[1],), p["fc.1.weight"], p["fc.1.bias"], 1e-5))
    if masks is not None:
        a = a * masks["head"]
    return F.linear(a, p["fc.4.weight"], p["fc.4.bias"]).squeeze(-1)


class AdamWState:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.lr, self.b1, self.b2, self.eps, self.wd, self.t = lr, betas[0], betas[1], eps, weight_decay, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        for k in params:
            g = grads[k]
            params[k].mul_(1 - self.lr * self.wd)
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / (bc2 ** 0.5)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def train_step(params, opt, x, y, masks=None, beta=2.0, max_norm=1.0, lengths=None):
    """One optimiser step in place on `params` (dict of leaf-less tensors).  Returns (loss, grad_norm)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = predictor_forward(leaf, x, masks, lengths)
    loss = F.smooth_l1_loss(out, y, beta=beta)
    loss.backward()
    grads = {k: leaf[k].grad for k in params}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    coef = min(1.0, max_norm / (total + 1e-6))
    grads = {k: g * coef for k, g in grads.items()}
    with torch.no_grad():
        opt.step(params, grads)
    return float(loss), total


def trajectory_dataset(segments, stop_radius=10, window_size=20, rng=None):
    """segments: dicts with positions [w, 2], concentrations [w], source_pos [2].  rng: a random.Random (the reference uses the
    global `random` module).  Returns (features [n, window, 1] f64, labels [n, 2] f64)."""
    import random as _random
    rng = rng or _random
    groups = {}
    for seg in segments:
        groups.setdefault(tuple(seg["source_pos"]), []).append(seg)
    chosen = rng.sample(list(groups.values()), min(1000, len(groups)))
    feats, labels = [], []
    for segs in chosen:
        seg = segs[0]
        conc = np.asarray(seg["concentrations"])
        if len(conc) < window_size:
            continue
        feats.append(conc[:window_size].reshape(-1, 1) / 100.0)
        labels.append([conc[window_size - 1] / 100.0, 0.0])
        feats.append(conc[-window_size:].reshape(-1, 1) / 100.0)
        near = np.linalg.norm(np.asarray(seg["positions"][-1]) - np.asarray(seg["source_pos"])) <= stop_radius
        labels.append([conc[-1] / 100.0, 1.0 if near else 0.0])
    return (np.stack(feats) if feats else np.zeros((0, window_size, 1))), np.asarray(labels, np.float64).reshape(-1, 2)


def peak_stop_forward(p, x):
    B, H = x.shape[0], p["lstm.weight_hh_l0"].shape[1]
    z = torch.zeros(B, H, dtype=x.dtype)
    _, hn, _ = po.lstm_layer_forward(x.transpose(0, 1), z, z, p["lstm.weight_ih_l0"], p["lstm.weight_hh_l0"],
                                     p["lstm.bias_ih_l0"], p["lstm.bias_hh_l0"], None)
    peak = F.linear(hn, p["fc_peak.weight"], p["fc_peak.bias"]).squeeze(-1)
    stop = torch.sigmoid(F.linear(hn, p["fc_stop.0.weight"], p["fc_stop.0.bias"])).squeeze(-1)
    return peak, stop


def peak_stop_train_step(params, opt, x, y, max_norm=1.0):
    """x [B, T, 1], y [B, 2] = (peak, stop) labels.  One step in place on `params`; returns (loss, grad_norm)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    peak, stop = peak_stop_forward(leaf, x)
    loss = F.mse_loss(peak, y[:, 0]) + F.binary_cross_entropy(stop, y[:, 1])
    loss.backward()
    grads = {k: leaf[k].grad for k in params}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    coef = min(1.0, max_norm / (total + 1e-6))
    grads = {k: g * coef for k, g in grads.items()}
    with torch.no_grad():
        opt.step(params, grads)
    return float(loss), total


class ReduceLROnPlateauOracle:
    def __init__(self, lr, factor=0.5, patience=5, threshold=1e-4, min_lr=0.0, eps=1e-8):
        self.lr, self.factor, self.patience, self.threshold, self.min_lr, self.eps = lr, factor, patience, threshold, min_lr, eps
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):        # mode 'min', threshold_mode 'rel'
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.lr * self.factor, self.min_lr)
            if self.lr - new > self.eps:
                self.lr = new
            self.bad = 0
        return self.lr
