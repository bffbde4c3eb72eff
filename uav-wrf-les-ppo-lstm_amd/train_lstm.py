"""train_lstm.py -- offline training of the LSTM stop predictor on the MI355X path.

Counterpart of the reference's PPOV2.0/train_lstm.py: SequenceDataset (:12-50) and the loop body of train_lstm()
(:60-99: ConcentrationThresholdPredictor, SmoothL1Loss(beta=2), AdamW(3e-4), clip_grad_norm_(1.0),
ReduceLROnPlateau(0.5, patience 5), 150 epochs, batch 64).  The three-layer LSTM runs through uav_lstm_fwd /
uav_lstm_bwd / uav_lstm_wgrad (one call per layer, dx handed down), the head through uav_gemm_f32 + uav_ln_relu(+_bwd),
the loss through uav_smooth_l1 and the update through uav_clip_adamw on one flat parameter buffer.  Dropout (0.3
between LSTM layers, 0.1 in the head) is applied as explicit masks drawn from a torch generator on the device, so a
training step is reproducible and comparable with the oracle given the same masks.  No CPU fallback.

The PPOV2.1 variant of the same stage (PPOV2.1/train_lstm.py) is here too: TrajectoryDataset (:11-74), and train() (:76-125)
as train_peak_and_stop(): PeakAndStopPredictor, MSELoss + BCELoss (uav_mse_bce), clip, AdamW(1e-3, wd 1e-4), plateau
scheduler (factor 0.1, patience 5), 100 epochs, batch 64, best checkpoint kept.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from config import TRAINING_SIZE
from evaluate_with_lstm import ConcentrationThresholdPredictor, PeakAndStopPredictor
from uavppo import ops

F32 = torch.float32


class SequenceDataset:
    """train_lstm.py:12-50: the last TRAINING_SIZE concentrations of every sequence that is long enough, scaled by one
    global min/max over all those windows; label = the episode's source concentration."""

    def __init__(self, sequences, source_concs, training_size=TRAINING_SIZE):
        wins = [np.asarray(s[-training_size:], np.float64) for s in sequences if len(s) >= training_size]
        self.labels = np.asarray([c for s, c in zip(sequences, source_concs) if len(s) >= training_size], np.float32)
        if wins:
            allv = np.concatenate(wins)
            self.data_min_, self.data_max_ = np.array([allv.min()]), np.array([allv.max()])
            rng = float(self.data_max_[0] - self.data_min_[0]) or 1.0
            self.windows = np.stack([((np.asarray(w, np.float32) - self.data_min_[0]) / rng) for w in wins]).astype(np.float32)
        else:
            self.data_min_ = self.data_max_ = np.array([np.nan])
            self.windows = np.zeros((0, training_size), np.float32)

    def __len__(self):
        return len(self.windows)

    def __getitem__(self, idx):
        return torch.from_numpy(self.windows[idx]), torch.tensor([self.labels[idx]])


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience) on a plain float (train_lstm.py:68-73)."""

    def __init__(self, lr, factor=0.5, patience=5, threshold=1e-4, min_lr=0.0, eps=1e-8):
        self.lr, self.factor, self.patience, self.threshold, self.min_lr, self.eps = lr, factor, patience, threshold, min_lr, eps
        self.best, self.num_bad_epochs = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new = max(self.lr * self.factor, self.min_lr)
            if self.lr - new > self.eps:
                self.lr = new
            self.num_bad_epochs = 0
        return self.lr


class PredictorTrainer:
    """Forward + backward + AdamW for a ConcentrationThresholdPredictor whose parameters are re-homed into one flat buffer."""

    P_LSTM, P_HEAD = 0.3, 0.1          # nn.LSTM(dropout=0.3), nn.Dropout(0.1): model.py:211,216

    def __init__(self, model: ConcentrationThresholdPredictor, lr=3e-4, weight_decay=0.01, seed=0):
        self.model, self.lr, self.wd, self.step_count = model, lr, weight_decay, 0
        dev = model.device
        names = [(f"lstm.{k}", v) for k, v in model.lstm.p.items()] + list(model.fc.items())
        total = sum(v.numel() for _, v in names)
        self.flat = torch.empty(total, dtype=F32, device=dev)
        self.grad = torch.zeros(total, dtype=F32, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.g = {}
        o = 0
        for name, v in names:                                   # parameters become views of the flat buffer
            n = v.numel()
            self.flat[o:o + n].copy_(v.reshape(-1))
            view = self.flat[o:o + n].view(v.shape)
            if name.startswith("lstm."):
                model.lstm.p[name[5:]] = view
            else:
                model.fc[name] = view
            self.g[name] = self.grad[o:o + n].view(v.shape)
            o += n
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        self.gnorm = torch.zeros(1, dtype=F32, device=dev)

    def draw_masks(self, B, T, train=True):
        H, dev = self.model.lstm.hidden_size, self.model.device
        if not train:
            return None
        m = lambda shape, p: (torch.rand(shape, generator=self.gen, device=dev) >= p).to(F32) / (1.0 - p)
        return {"l0": m((B, T, H), self.P_LSTM), "l1": m((B, T, H), self.P_LSTM), "head": m((B, 64), self.P_HEAD)}

    def train_step(self, x, y, masks=None, beta=2.0, max_norm=1.0, lengths=None):
        """x [B, T, 1], y [B] on the device; masks from draw_masks() or None (eval-mode step).  lengths: None (every sequence
        has T steps -- all SequenceDataset produces) or B ints in 1..T for zero-padded ragged sequences, the
        ConcentrationThresholdPredictor.forward(x, lengths) of model.py:229-240: sequence i ends at step lengths[i] - 1, that
        output feeds the head and only that step receives the head's gradient (the padded steps run and are ignored: an LSTM is
        causal, so this equals pack_padded_sequence).  Returns the loss (f64[1] tensor)."""
        m, p, fc, g = self.model, self.model.lstm.p, self.model.fc, self.g
        B, T, _ = x.shape
        H = m.lstm.hidden_size
        rows = torch.arange(B, device=x.device)
        if lengths is None:
            last = torch.full((B,), T - 1, dtype=torch.long, device=x.device)
        else:
            last = torch.as_tensor(np.asarray(lengths), dtype=torch.long, device=x.device) - 1
            if last.numel() != B or int(last.min()) < 0 or int(last.max()) >= T:
                raise ValueError(f"lengths: {B} values in 1..{T} expected")
        z0 = torch.zeros(B, H, dtype=F32, device=x.device)
        # ---- forward, keeping what BPTT needs
        xs, ys, stashes = [x.contiguous()], [], []
        for l in range(3):
            yl, _, _, st = ops.lstm_fwd(xs[l], None, z0, z0, p[f"weight_ih_l{l}"], p[f"weight_hh_l{l}"], p[f"bias_ih_l{l}"],
                                        p[f"bias_hh_l{l}"])
            ys.append(yl)
            stashes.append(st)
            if l < 2:
                xs.append(yl * masks[f"l{l}"] if masks is not None else yl)
        h = ys[2][rows, last].contiguous()
        z = ops.gemm(h, fc["fc.0.weight"], trans_b=True, bias=fc["fc.0.bias"])
        a, rstd = ops.ln_relu(z, fc["fc.1.weight"], fc["fc.1.bias"], want_stats=True)      # z now holds xhat
        ad = a * masks["head"] if masks is not None else a
        out = ops.gemm(ad, fc["fc.4.weight"], trans_b=True, bias=fc["fc.4.bias"]).reshape(-1)
        loss, dout = ops.smooth_l1(out, y.contiguous(), beta)
        # ---- backward: head
        dout2 = dout.view(B, 1)
        ops.gemm(dout2, ad, trans_a=True, out=g["fc.4.weight"])
        g["fc.4.bias"].copy_(dout.sum(0, keepdim=True))
        da = ops.gemm(dout2, fc["fc.4.weight"])
        if masks is not None:
            da = da * masks["head"]
        dz, dgam, dbet = ops.ln_relu_bwd(da.contiguous(), z, rstd, fc["fc.1.weight"], fc["fc.1.bias"])
        g["fc.1.weight"].copy_(dgam)
        g["fc.1.bias"].copy_(dbet)
        ops.gemm(dz, h, trans_a=True, out=g["fc.0.weight"])
        ops.colsum(dz, out=g["fc.0.bias"])
        dh = ops.gemm(dz, fc["fc.0.weight"])
        # ---- backward: LSTM stack, top layer first; only the last step of each sequence receives the head's gradient
        dy = torch.zeros(B, T, H, dtype=F32, device=x.device)
        dy[rows, last] = dh
        for l in (2, 1, 0):
            r = ops.lstm_bwd(xs[l], None, stashes[l], p[f"weight_ih_l{l}"], p[f"weight_hh_l{l}"], ys[l], z0, dy=dy,
                             need_dx=(l > 0), dw_ih=g[f"lstm.weight_ih_l{l}"], dw_hh=g[f"lstm.weight_hh_l{l}"],
                             db=g[f"lstm.bias_ih_l{l}"], want_dstate=False, db_hh=g[f"lstm.bias_hh_l{l}"])
            if l > 0:
                dy = r["dx"] * masks[f"l{l - 1}"] if masks is not None else r["dx"]
                dy = dy.contiguous()
        # ---- clip + AdamW
        self.step_count += 1
        ops.clip_adamw(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, weight_decay=self.wd,
                       max_norm=max_norm, gnorm_out=self.gnorm)
        return loss


def train_lstm(sequences=None, source_concs=None, epochs=150, batch_size=64, device="cuda", seed=0, nc_path="training_data.nc",
               model_dir="model"):
    """train_lstm() of the reference (:52-99).  sequences / source_concs default to load_raw_sequences(nc_path)."""
    if sequences is None:
        from data_loader import load_raw_sequences
        sequences, source_concs = load_raw_sequences(nc_path)
    pairs = [(s, c) for s, c in zip(sequences, source_concs) if len(s) >= 10]
    sequences, source_concs = zip(*pairs) if pairs else ([], [])
    ds = SequenceDataset(sequences, source_concs, TRAINING_SIZE)
    model = ConcentrationThresholdPredictor(input_size=1, hidden_size=128, device=device, seed=seed)
    tr = PredictorTrainer(model, lr=3e-4, seed=seed)
    sched = ReduceLROnPlateau(3e-4, factor=0.5, patience=5)
    X = torch.from_numpy(ds.windows).to(device)
    Y = torch.from_numpy(ds.labels).to(device)
    perm_gen = torch.Generator().manual_seed(seed)
    history = []
    for epoch in range(epochs):
        order = torch.randperm(len(ds), generator=perm_gen).to(device)          # DataLoader(shuffle=True)
        total, nb = 0.0, 0
        for s in range(0, len(ds), batch_size):
            idx = order[s:s + batch_size]
            xb, yb = X[idx][:, :, None].contiguous(), Y[idx].contiguous()
            loss = tr.train_step(xb, yb, tr.draw_masks(len(idx), xb.shape[1]))
            total += float(loss.item())
            nb += 1
        avg = total / max(nb, 1)
        tr.lr = sched.step(avg)
        history.append(avg)
        print(f"Epoch {epoch + 1}, Loss: {avg:.4f}")
    if model_dir:
        os.makedirs(model_dir, exist_ok=True)
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(model_dir, "lstm_threshold_predictor.pth"))
        np.save(os.path.join(model_dir, "scaler_params.npy"), ds.data_min_)        # as the reference does (:98)
    return model, history


class TrajectoryDataset:
    """PPOV2.1/train_lstm.py:11-74.  One negative (first window, stop label 0) and one positive (last window, stop label =
    final position within stop_radius of the source) sample from the first segment of up to 1000 randomly chosen episodes;
    concentrations / 100.  Episodes are drawn with the `random` module, as in the reference (seed it for repeatability)."""

    def __init__(self, segments, stop_radius=10, window_size=20, rng=None):
        import random
        from config import GRID_SIZE
        self.grid_size, self.stop_radius, self.window_size, self.segments = GRID_SIZE, stop_radius, window_size, segments
        self.max_sigma = max([seg.get("sigma", 15.0) for seg in segments]) if segments else 50.0
        self.max_peak = max([seg["concentrations"][-1] for seg in segments]) if segments else 100.0
        rng = rng or random
        episodes = {}
        for seg in segments:
            episodes.setdefault(tuple(seg["source_pos"]), []).append(seg)
        feats, labels = [], []
        for segs in rng.sample(list(episodes.values()), min(1000, len(episodes))):
            seg = segs[0]
            conc = np.asarray(seg["concentrations"])
            if len(conc) < window_size:
                continue
            feats.append(conc[:window_size].reshape(-1, 1) / 100.0)
            labels.append([conc[window_size - 1] / 100.0, 0.0])
            feats.append(conc[-window_size:].reshape(-1, 1) / 100.0)
            near = np.linalg.norm(np.asarray(seg["positions"][-1]) - np.asarray(seg["source_pos"])) <= stop_radius
            labels.append([conc[-1] / 100.0, 1.0 if near else 0.0])
        self.features, self.labels = feats, np.array(labels)
        print(f"samples collected: {len(labels)} (positive + negative)")

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        return torch.as_tensor(self.features[idx], dtype=F32), torch.as_tensor(self.labels[idx], dtype=F32)


class PeakStopTrainer:
    """Forward + backward + AdamW for a PeakAndStopPredictor (single-layer LSTM, two linear heads sharing one GEMM)."""

    def __init__(self, model: PeakAndStopPredictor, lr=1e-3, weight_decay=1e-4):
        self.model, self.lr, self.wd, self.step_count = model, lr, weight_decay, 0
        dev = model.device
        names = [(f"lstm.{k}", v) for k, v in model.lstm.p.items()] + [("heads_w", model.heads_w), ("heads_b", model.heads_b)]
        total = sum(v.numel() for _, v in names)
        self.flat = torch.empty(total, dtype=F32, device=dev)
        self.grad = torch.zeros(total, dtype=F32, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.g, o = {}, 0
        for name, v in names:
            n = v.numel()
            self.flat[o:o + n].copy_(v.reshape(-1))
            view = self.flat[o:o + n].view(v.shape)
            if name.startswith("lstm."):
                model.lstm.p[name[5:]] = view
            else:
                setattr(model, name, view)
            self.g[name] = self.grad[o:o + n].view(v.shape)
            o += n
        self.gnorm = torch.zeros(1, dtype=F32, device=dev)

    def train_step(self, x, y, max_norm=1.0):
        """x [B, T, 1], y [B, 2] = (peak, stop) labels on the device.  Returns the loss (f64[1] tensor)."""
        m, p, g = self.model, self.model.lstm.p, self.g
        B, T, _ = x.shape
        H = m.lstm.hidden_size
        z0 = torch.zeros(B, H, dtype=F32, device=x.device)
        x = x.contiguous()
        yl, _, _, st = ops.lstm_fwd(x, None, z0, z0, p["weight_ih_l0"], p["weight_hh_l0"], p["bias_ih_l0"], p["bias_hh_l0"])
        h = yl[:, T - 1].contiguous()
        out = ops.gemm(h, m.heads_w, trans_b=True, bias=m.heads_b)              # [B, 2] = (peak, stop logit)
        loss, dout = ops.mse_bce(out, y.contiguous())
        ops.gemm(dout, h, trans_a=True, out=g["heads_w"])
        ops.colsum(dout, out=g["heads_b"])
        dy = torch.zeros(B, T, H, dtype=F32, device=x.device)
        dy[:, T - 1] = ops.gemm(dout, m.heads_w)
        ops.lstm_bwd(x, None, st, p["weight_ih_l0"], p["weight_hh_l0"], yl, z0, dy=dy, need_dx=False,
                     dw_ih=g["lstm.weight_ih_l0"], dw_hh=g["lstm.weight_hh_l0"], db=g["lstm.bias_ih_l0"], want_dstate=False,
                     db_hh=g["lstm.bias_hh_l0"])
        self.step_count += 1
        ops.clip_adamw(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, weight_decay=self.wd,
                       max_norm=max_norm, gnorm_out=self.gnorm)
        return loss


def train_peak_and_stop(segments=None, epochs=100, batch_size=64, device="cuda", seed=0, nc_path="training_data.nc",
                        model_dir="model"):
    """train() of PPOV2.1/train_lstm.py:76-125.  segments default to load_trajectory_segments(nc_path, tail_steps=60)."""
    if segments is None:
        from data_loader import load_trajectory_segments
        segments = load_trajectory_segments(nc_path, tail_steps=60)
    ds = TrajectoryDataset(segments, window_size=20)
    model = PeakAndStopPredictor(input_dim=1, device=device, seed=seed)
    tr = PeakStopTrainer(model, lr=1e-3, weight_decay=1e-4)
    sched = ReduceLROnPlateau(1e-3, factor=0.1, patience=5)                       # torch's defaults, as the reference (:104)
    X = torch.as_tensor(np.stack(ds.features), dtype=F32).to(device)
    Y = torch.as_tensor(ds.labels, dtype=F32).to(device)
    perm_gen = torch.Generator().manual_seed(seed)
    best, history = float("inf"), []
    for epoch in range(epochs):
        order = torch.randperm(len(ds), generator=perm_gen).to(device)
        total, nb = 0.0, 0
        for s in range(0, len(ds), batch_size):
            idx = order[s:s + batch_size]
            total += float(tr.train_step(X[idx].contiguous(), Y[idx].contiguous()).item())
            nb += 1
        avg = total / max(nb, 1)
        tr.lr = sched.step(avg)
        history.append(avg)
        if avg < best:
            best = avg
            if model_dir:
                os.makedirs(model_dir, exist_ok=True)
                torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(model_dir, "best_peak_and_stop.pth"))
        print(f"Epoch {epoch + 1:03d} | Loss: {avg:.4f} | LR: {tr.lr:.2e}")
    return model, history


if __name__ == "__main__":
    train_lstm()
