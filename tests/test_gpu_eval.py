"""N2: greedy evaluation + LSTM stop controller on the GPU vs the evaluation oracle (itself pinned to the
reference's classes by tests/test_oracle_eval.py).  -m gpu."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo
from oracle import ppo_oracle as po
from oracle.env_oracle import FieldBank, OracleVecEnv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_v20.npz")


@pytest.fixture(scope="module")
def ev():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uav-wrf-les-ppo-lstm_amd")
    if root not in sys.path:
        sys.path.insert(0, root)
    import evaluate_with_lstm as m
    return m


def _cpu_sd(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("H,B,T", [(128, 33, 10), (32, 7, 10), (64, 5, 23)])
def test_threshold_predictor_matches_oracle(ev, H, B, T):
    net = ev.ConcentrationThresholdPredictor(hidden_size=H, device=DEV, seed=3)
    for k in ("fc.4.bias",):
        net.fc[k].fill_(0.7)
    x = torch.rand(B, T, 1, device=DEV)
    want = eo.ThresholdPredictorOracle(_cpu_sd(net))(x.cpu().numpy())
    got = net(x)
    assert torch.allclose(got.cpu(), want, rtol=2e-5, atol=2e-5)
    # pack_padded_sequence semantics: the output of the last VALID step of every row
    lengths = [T - (i % 3) for i in range(B)]
    got_l = net(x, lengths=lengths).cpu()
    for i in (0, 1, 2, B - 1):
        w_i = eo.ThresholdPredictorOracle(_cpu_sd(net))(x[i:i + 1, :lengths[i]].cpu().numpy())
        assert abs(float(got_l[i]) - float(w_i)) < 5e-5


def test_threshold_predictor_loads_the_reference_golden(ev):
    g = np.load(GOLD, allow_pickle=False)
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    net = ev.ConcentrationThresholdPredictor(hidden_size=32, device=DEV)
    net.load_state_dict(sd)
    y = net(torch.from_numpy(g["pred_x"]).to(DEV)).cpu().numpy()
    assert np.allclose(y, g["pred_y"], rtol=2e-5, atol=2e-4)        # outputs are ~50 (concentration units)


def test_threshold_controller_matches_reference_golden(ev):
    """The reference's ThresholdController traces (thresholds, stop steps) replayed through the vectorised controller:
    8 recorded trajectories = 8 parallel 'environments'."""
    g = np.load(GOLD, allow_pickle=False)
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    net = ev.ConcentrationThresholdPredictor(hidden_size=32, device=DEV)
    net.load_state_dict(sd)
    traj, thr, stop_at = g["traj"], g["thresholds"], g["stop_at"]
    E, L = traj.shape
    ctl = ev.ThresholdController(net, (g["scaler_params"].min(), g["scaler_params"].max()), E, device=DEV)
    stop = np.full(E, -1)
    for step in range(1, L + 1):
        cur = torch.from_numpy(traj[:, step - 1]).to(DEV)
        ctl.push(cur)
        if step % 10 == 0:
            ctl.update_threshold()
            got = ctl.current_threshold.cpu().numpy()
            for e in range(E):
                if stop[e] < 0 and not np.isnan(thr[e][step // 10 - 1]):
                    assert abs(got[e] - thr[e][step // 10 - 1]) < 2e-3 * max(1.0, abs(thr[e][step // 10 - 1])), (e, step)
        s = ctl.should_stop(cur, step).cpu().numpy()
        for e in range(E):
            if stop[e] < 0 and s[e]:
                stop[e] = step
    assert np.array_equal(stop, stop_at)


def test_peak_and_stop_predictor_matches_oracle(ev):
    net = ev.PeakAndStopPredictor(device=DEV, seed=5)
    x = torch.rand(19, 20, 1, device=DEV)
    peak_w, stop_w = eo.PeakStopPredictorOracle(_cpu_sd(net))(x.cpu().numpy())
    peak, stop = net(x)
    assert torch.allclose(peak.cpu(), peak_w, atol=2e-5) and torch.allclose(stop.cpu(), stop_w, atol=2e-5)
    peak2, _ = net(x[:, :, 0])                       # 2-D input is unsqueezed like the reference's forward
    assert torch.equal(peak, peak2)


def test_vectorised_greedy_evaluation_matches_oracle_episodes(ev):
    """N greedy episodes with the PPOV2.0 stop controller, materialised fields + injected noise: steps, early-stop
    flags and final deviations equal N sequential oracle episodes (env bit-exact; decisions identical)."""
    from uavppo.policy import MLPActorCritic
    from uavppo.vec_env import VecMethaneEnv
    N, LIM = 12, 120
    bank = FieldBank.from_seed(N, "v2.0", seed=9)
    env = VecMethaneEnv(N, "v2.0", DEV, seed=3, bank=bank.interleaved(), bank_sources=bank.sources)
    pol = MLPActorCritic(6, 5, device=DEV, seed=8)
    pol.views["head.weight"][:5].mul_(40.0)              # a decisive (non-uniform) greedy policy
    pred = ev.ConcentrationThresholdPredictor(hidden_size=64, device=DEV, seed=4)
    pred.fc["fc.4.bias"].fill_(18.0)                      # thresholds inside the plume's concentration range
    pred.fc["fc.4.weight"].mul_(6.0)
    scaler = (0.0, 100.0)
    rng = np.random.RandomState(1)
    noise = rng.randn(LIM, N, 2)
    ctl = ev.ThresholdController(pred, scaler, N, device=DEV)
    got = ev.evaluate(lambda o: pol.heads(o.contiguous())[:, :5], env, ctl, noise=torch.from_numpy(noise).to(DEV), max_steps=LIM)

    # oracle: one env at a time
    p = {k: v.detach().cpu() for k, v in pol.named_views().items()}
    onet = eo.ThresholdPredictorOracle(_cpu_sd(pred))
    ora = OracleVecEnv(N, bank, "v2.0", radius=50.0)
    ora.reset()
    steps, stopped, devs = [], [], []
    for i, e in enumerate(ora.envs):
        octl = eo.ThresholdControllerOracle(onet, np.array(scaler))
        state = e.obs()
        traj, t, done, st = [], 0, False, False
        while not done and t < LIM:
            with torch.no_grad():
                probs, _, _ = po.mlp_forward(p, torch.from_numpy(state)[None])
            a = int(torch.argmax(probs))
            state, _, done, _reached, _info = e.step(a, noise[t, i])
            cur = float(state[2]) * 100.0
            traj.append(cur)
            t += 1
            if t % 10 == 0:
                octl.update_threshold(traj)
            if octl.should_stop(cur, t):
                st, done = True, True
        steps.append(t)
        stopped.append(st)
        devs.append(float(np.linalg.norm(np.asarray(e.pos, np.float64) - np.asarray(e.source, np.float64))))
    assert np.array_equal(got["steps"], np.asarray(steps)), (got["steps"], steps)
    assert np.array_equal(got["stopped_early"], np.asarray(stopped))
    assert np.allclose(got["deviations"], np.asarray(devs), atol=2e-3)
    assert 0 < np.sum(stopped) < N or np.ptp(steps) > 0          # the scenario is not degenerate


def test_vectorised_greedy_evaluation_v21_stop_rule_matches_oracle(ev):
    """PPOV2.1 rule (evaluate_with_lstm.py:69-77): once 20 concentrations are in the trajectory, stop when the
    PeakAndStopPredictor's stop probability exceeds 0.8.  N parallel episodes == N sequential oracle episodes."""
    from uavppo.policy import MLPActorCritic
    from uavppo.vec_env import VecMethaneEnv
    N, LIM = 10, 90
    bank = FieldBank.from_seed(N, "v2.1", seed=21)
    env = VecMethaneEnv(N, "v2.1", DEV, seed=5, bank=bank.interleaved(), bank_sources=bank.sources)
    pol = MLPActorCritic(6, 5, device=DEV, seed=12)
    pol.views["head.weight"][:5].mul_(40.0)
    pred = ev.PeakAndStopPredictor(device=DEV, seed=6)
    pred.heads_w[1].mul_(12.0)                 # make the stop head decisive one way or the other
    rng = np.random.RandomState(4)
    noise = rng.randn(LIM, N, 2)
    got = ev.evaluate(lambda o: pol.heads(o.contiguous())[:, :5], env, None, peak_stop=pred,
                      noise=torch.from_numpy(noise).to(DEV), max_steps=LIM, success_distance=50)
    p = {k: v.detach().cpu() for k, v in pol.named_views().items()}
    onet = eo.PeakStopPredictorOracle(_cpu_sd(pred))
    ora = OracleVecEnv(N, bank, "v2.1", radius=50.0)
    ora.reset()
    steps, stopped = [], []
    for i, e in enumerate(ora.envs):
        state, traj, t, done, st = e.obs(), [], 0, False, False
        while not done and t < LIM:
            with torch.no_grad():
                probs, _, _ = po.mlp_forward(p, torch.from_numpy(state)[None])
            state, _, done, _r, _i = e.step(int(torch.argmax(probs)), noise[t, i])
            traj.append(float(state[2]) * 100.0)
            t += 1
            hit, _peak = eo.stop_rule_v21(onet, traj)
            if hit:
                st, done = True, True
        steps.append(t)
        stopped.append(st)
    assert np.array_equal(got["steps"], np.asarray(steps)), (got["steps"], steps)
    assert np.array_equal(got["stopped_early"], np.asarray(stopped))
    assert np.isnan(got["peak_pred"][~np.asarray(stopped)]).all() and np.isfinite(got["peak_pred"][np.asarray(stopped)]).all()
