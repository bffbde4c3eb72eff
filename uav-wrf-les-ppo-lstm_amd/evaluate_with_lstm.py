"""evaluate_with_lstm.py -- greedy evaluation with an LSTM stop controller on the MI355X path.

Counterpart of the reference's PPOV2.0/evaluate_with_lstm.py (ThresholdController, :10-37; episode loop :67-101)
and PPOV2.1/evaluate_with_lstm.py (PeakAndStopPredictor :11-27; stop rule :69-77), vectorised: the reference's
1000 sequential episodes become N environments stepped together (one episode each), the greedy policy forward is
one uav_mlp_fwd / LSTM step per time step, and the stop predictors run through uav_lstm_fwd + uav_gemm_f32 +
uav_ln_relu.  Same class names, state_dict keys, metrics keys and decision rules as the reference; per-episode
Python scalars become device tensors of length N.  No CPU fallback: everything goes through uavppo.ops.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from config import EVALUATE_SIZE, SUCCESS_DISTANCE_THRESHOLD
from uavppo import ops

F32 = torch.float32


def _xavier(shape, gen):
    w = torch.empty(shape)
    torch.nn.init.xavier_uniform_(w, generator=gen)
    return w


class _DeviceLSTMStack:
    """nn.LSTM(input, hidden, num_layers, batch_first=True) in eval mode (no dropout), zero initial state."""

    def __init__(self, input_size, hidden_size, num_layers, device, gen, xavier):
        self.input_size, self.hidden_size, self.num_layers, self.device = input_size, hidden_size, num_layers, device
        self.p = {}
        k = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            i = input_size if l == 0 else hidden_size
            for name, shape in ((f"weight_ih_l{l}", (4 * hidden_size, i)), (f"weight_hh_l{l}", (4 * hidden_size, hidden_size)),
                                (f"bias_ih_l{l}", (4 * hidden_size,)), (f"bias_hh_l{l}", (4 * hidden_size,))):
                if xavier:      # ConcentrationThresholdPredictor._init_weights, model.py:222-227
                    t = _xavier(shape, gen) if len(shape) > 1 else torch.zeros(shape)
                else:           # nn.LSTM default
                    t = (torch.rand(shape, generator=gen) * 2 - 1) * k
                self.p[name] = t.to(device=device, dtype=F32).contiguous()

    def last_output(self, x, lengths=None):
        """x [B, T, I] -> top layer output at the last valid step of every row [B, H]."""
        B, T, _ = x.shape
        H = self.hidden_size
        z = torch.zeros(B, H, dtype=F32, device=x.device)
        seq = x.contiguous()
        for l in range(self.num_layers):
            p = self.p
            seq, hn, cn, _ = ops.lstm_fwd(seq, None, z, z, p[f"weight_ih_l{l}"], p[f"weight_hh_l{l}"], p[f"bias_ih_l{l}"],
                                          p[f"bias_hh_l{l}"], want_stash=False)
        if lengths is None:
            return seq[:, T - 1].contiguous()
        idx = torch.as_tensor(lengths, device=x.device, dtype=torch.long) - 1          # pack_padded_sequence semantics
        return seq[torch.arange(B, device=x.device), idx].contiguous()


class ConcentrationThresholdPredictor:
    """PPOV2.0/model.py:203-240, inference: 3-layer LSTM(1 -> hidden) -> Linear(hidden, 64) -> LayerNorm(64) -> ReLU ->
    Linear(64, 1).  state_dict keys as the reference's nn.Module (lstm.*, fc.0.*, fc.1.*, fc.4.*)."""

    def __init__(self, input_size=1, hidden_size=128, device="cuda", seed=None):
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.device = torch.device(device)
        self.lstm = _DeviceLSTMStack(input_size, hidden_size, 3, self.device, gen, xavier=True)
        d = dict(device=self.device, dtype=F32)
        self.fc = {"fc.0.weight": _xavier((64, hidden_size), gen).to(**d), "fc.0.bias": torch.zeros(64, **d),
                   "fc.1.weight": torch.ones(64, **d), "fc.1.bias": torch.zeros(64, **d),
                   "fc.4.weight": _xavier((1, 64), gen).to(**d), "fc.4.bias": torch.zeros(1, **d)}

    def state_dict(self):
        sd = {f"lstm.{k}": v.clone() for k, v in self.lstm.p.items()}
        sd.update({k: v.clone() for k, v in self.fc.items()})
        return sd

    def load_state_dict(self, sd):
        for k in self.lstm.p:
            self.lstm.p[k].copy_(torch.as_tensor(np.asarray(sd[f"lstm.{k}"]), dtype=F32).reshape(self.lstm.p[k].shape))
        for k in self.fc:
            self.fc[k].copy_(torch.as_tensor(np.asarray(sd[k]), dtype=F32).reshape(self.fc[k].shape))

    def eval(self):
        return self

    def __call__(self, x, lengths=None):
        h = self.lstm.last_output(x, lengths)
        z = ops.gemm(h, self.fc["fc.0.weight"], trans_b=True, bias=self.fc["fc.0.bias"])
        a = ops.ln_relu(z, self.fc["fc.1.weight"], self.fc["fc.1.bias"])
        return ops.gemm(a, self.fc["fc.4.weight"], trans_b=True, bias=self.fc["fc.4.bias"]).reshape(-1)


class PeakAndStopPredictor:
    """PPOV2.1/evaluate_with_lstm.py:11-27: LSTM(1 -> hidden) -> h_n -> (Linear -> peak, Linear + Sigmoid -> stop_prob)."""

    def __init__(self, input_dim=1, hidden_dim=32, num_layers=1, device="cuda", seed=None):
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.device = torch.device(device)
        self.lstm = _DeviceLSTMStack(input_dim, hidden_dim, num_layers, self.device, gen, xavier=False)
        k = 1.0 / math.sqrt(hidden_dim)
        d = dict(device=self.device, dtype=F32)
        u = lambda shape: ((torch.rand(shape, generator=gen) * 2 - 1) * k).to(**d)
        # the two heads share one GEMM: rows = (peak, stop)
        self.heads_w, self.heads_b = torch.cat([u((1, hidden_dim)), u((1, hidden_dim))]).contiguous(), torch.cat([u((1,)), u((1,))])

    def state_dict(self):
        sd = {f"lstm.{k}": v.clone() for k, v in self.lstm.p.items()}
        sd.update({"fc_peak.weight": self.heads_w[0:1].clone(), "fc_peak.bias": self.heads_b[0:1].clone(),
                   "fc_stop.0.weight": self.heads_w[1:2].clone(), "fc_stop.0.bias": self.heads_b[1:2].clone()})
        return sd

    def load_state_dict(self, sd):
        for k in self.lstm.p:
            self.lstm.p[k].copy_(torch.as_tensor(np.asarray(sd[f"lstm.{k}"]), dtype=F32).reshape(self.lstm.p[k].shape))
        t = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=F32).to(self.device)
        self.heads_w.copy_(torch.cat([t("fc_peak.weight").reshape(1, -1), t("fc_stop.0.weight").reshape(1, -1)]))
        self.heads_b.copy_(torch.cat([t("fc_peak.bias").reshape(1), t("fc_stop.0.bias").reshape(1)]))

    def eval(self):
        return self

    def __call__(self, x):
        if x.dim() == 2:
            x = x.unsqueeze(-1)
        h = self.lstm.last_output(x)
        out = ops.gemm(h, self.heads_w, trans_b=True, bias=self.heads_b)
        return out[:, 0], torch.sigmoid(out[:, 1])


class ThresholdController:
    """PPOV2.0/evaluate_with_lstm.py:10-37 for N environments at once.  `scaler` is anything with data_min_/data_max_
    (sklearn's MinMaxScaler) or a (min, max) pair.  The reference's per-episode `conc_buffer` and `trajectory[-window:]`
    are the same last-`window_size` concentrations, kept here as one [N, window] device tensor in time order."""

    def __init__(self, model, scaler, num_envs, window_size=EVALUATE_SIZE, device="cuda"):
        self.model, self.window_size, self.N = model, int(window_size), int(num_envs)
        self.device = torch.device(device)
        self.min_activate_steps = 2 * self.window_size
        lo, hi = ((float(scaler.data_min_[0]), float(scaler.data_max_[0])) if hasattr(scaler, "data_min_")
                  else (float(scaler[0]), float(scaler[1])))
        self.lo, self.scale = lo, (hi - lo) if hi != lo else 1.0
        self.reset()

    def reset(self):
        d = self.device
        self.window = torch.zeros(self.N, self.window_size, dtype=torch.float64, device=d)
        self.count = torch.zeros(self.N, dtype=torch.int64, device=d)
        self.current_threshold = torch.full((self.N,), float("nan"), dtype=torch.float64, device=d)   # NaN = None

    def push(self, current_conc):
        self.window = torch.roll(self.window, -1, dims=1)
        self.window[:, -1] = current_conc
        self.count += 1

    def update_threshold(self, active=None):
        """every 10th step of the loop (evaluate_with_lstm.py:87-88): envs whose trajectory is long enough"""
        ok = self.count >= max(self.window_size, self.min_activate_steps)
        if active is not None:
            ok &= active
        scaled = ((self.window - self.lo) / self.scale).to(F32).reshape(self.N, self.window_size, 1)
        pred = self.model(scaled, lengths=None).to(torch.float64) * 0.95
        self.current_threshold = torch.where(ok, pred, self.current_threshold)

    def should_stop(self, current_conc, step_count):
        w = self.window_size
        n = torch.clamp(self.count, max=w).to(torch.float64)
        valid = (torch.arange(w, device=self.device)[None, :] >= (w - torch.clamp(self.count, max=w))[:, None])
        mean = (self.window * valid).sum(1) / torch.clamp(n, min=1.0)
        has = ~torch.isnan(self.current_threshold)
        thr = torch.nan_to_num(self.current_threshold, nan=float("inf"))
        return (step_count >= self.min_activate_steps) & has & ((current_conc >= thr) | (mean >= thr))


@torch.no_grad()
def evaluate(policy_probs, env, controller=None, peak_stop=None, window_size_v21=20, noise=None, max_steps=None,
             success_distance=SUCCESS_DISTANCE_THRESHOLD):
    """One greedy episode per environment of `env` (a uavppo VecMethaneEnv), all N together.

    policy_probs(obs [N, obs_dim]) -> probs or logits [N, 5] (argmax is taken);
    controller: ThresholdController (PPOV2.0 rule) or None; peak_stop: PeakAndStopPredictor (PPOV2.1 rule) or None;
    noise: optional f64 [steps, N, 2] (parity tests).  Returns the reference's metrics dict (deviations, steps, success,
    stopped_early [, peak_pred]) as numpy arrays of length N."""
    N, dev = env.num_envs, env.device
    obs = env.reset()
    _, src, _, _ = env.peek()
    src = src.clone()
    active = torch.ones(N, dtype=torch.bool, device=dev)
    steps = torch.zeros(N, dtype=torch.int64, device=dev)
    stopped = torch.zeros(N, dtype=torch.bool, device=dev)
    final_pos = torch.zeros(N, 2, dtype=torch.float64, device=dev)
    peak_pred = torch.full((N,), float("nan"), dtype=torch.float64, device=dev)
    if controller is not None:
        controller.reset()
    traj = torch.zeros(N, window_size_v21, dtype=torch.float64, device=dev) if peak_stop is not None else None
    limit = max_steps or env.max_steps
    for t in range(1, limit + 1):
        act = torch.argmax(policy_probs(obs), dim=1).to(torch.int32)
        obs, _, done, _ = env.step(act, None if noise is None else noise[t - 1])
        done_b = done > 0.5
        cur = torch.where(done_b, env.term_obs[:, 2], obs[:, 2]).to(torch.float64) * 100.0      # conc_field at the agent
        pos_now, _, _, _ = env.peek()
        pos_end = torch.where(done_b[:, None], env.term_obs[:, :2].to(torch.float64) * 500.0, pos_now.to(torch.float64))
        stop_now = torch.zeros(N, dtype=torch.bool, device=dev)
        if controller is not None:
            controller.push(cur)
            if t % 10 == 0:
                controller.update_threshold(active)
            stop_now |= controller.should_stop(cur, t)
        if peak_stop is not None:
            traj = torch.roll(traj, -1, dims=1)
            traj[:, -1] = cur
            if t >= window_size_v21:
                peak, prob = peak_stop((traj / 100.0).to(F32))
                hit = prob > 0.8
                peak_pred = torch.where(hit & active, peak.to(torch.float64), peak_pred)       # recorded whenever the LSTM stopped it (:85-87)
                stop_now |= hit
        ended = active & (done_b | stop_now)
        steps = torch.where(ended, torch.full_like(steps, t), steps)
        stopped |= ended & stop_now          # the reference sets the flag whenever the controller fires on the last step
        final_pos = torch.where(ended[:, None], pos_end, final_pos)
        active &= ~ended
        if t % 16 == 0 and not bool(active.any()):
            break
    # episodes cut off by `max_steps`
    if bool(active.any()):
        pos_now, _, _, _ = env.peek()
        final_pos = torch.where(active[:, None], pos_now.to(torch.float64), final_pos)
        steps = torch.where(active, torch.full_like(steps, limit), steps)
    deviation = torch.linalg.norm(final_pos - src, dim=1)
    out = {"deviations": deviation.cpu().numpy(), "steps": steps.cpu().numpy(),
           "success": (deviation <= success_distance).cpu().numpy(), "stopped_early": stopped.cpu().numpy()}
    if peak_stop is not None:
        out["peak_pred"] = peak_pred.cpu().numpy()
    return out


def main(num_envs=1000, model_dir="model", device="cuda"):
    """The reference's main() (evaluate_with_lstm.py:39-134) with its 1000 episodes run as 1000 parallel environments."""
    from model import PPOActorCritic
    from uavppo.vec_env import VecMethaneEnv
    ppo_model = PPOActorCritic(6, 5, device=device)
    lstm_model = ConcentrationThresholdPredictor(device=device)
    try:
        ppo_model.load_state_dict(torch.load(os.path.join(model_dir, "ppo_successful_models.pth"), map_location="cpu"))
        lstm_model.load_state_dict(torch.load(os.path.join(model_dir, "lstm_threshold_predictor.pth"), map_location="cpu"))
        scaler_params = np.load(os.path.join(model_dir, "scaler_params.npy"))
    except FileNotFoundError as e:
        print(f"model files missing: {e}")
        return None
    env = VecMethaneEnv(num_envs, "v2.0", device)
    controller = ThresholdController(lstm_model, (scaler_params.min(), scaler_params.max()), num_envs, device=device)
    metrics = evaluate(lambda o: ppo_model.core.heads(o)[:, :5], env, controller)     # argmax of logits == argmax of probs
    ok = metrics["success"]
    print("===== validation =====")
    print(f"mean deviation: {metrics['deviations'].mean():.2f} +- {metrics['deviations'].std():.2f} px")
    if ok.any():
        print(f"mean deviation of successes: {metrics['deviations'][ok].mean():.2f} +- {metrics['deviations'][ok].std():.2f} px")
    print(f"success rate: {ok.mean() * 100:.1f}%  early-stop rate: {metrics['stopped_early'].mean() * 100:.1f}%  "
          f"mean steps: {metrics['steps'].mean():.1f}")
    os.makedirs("results", exist_ok=True)
    np.savez("results/validation_metrics.npz", **metrics)
    return metrics


if __name__ == "__main__":
    main()
