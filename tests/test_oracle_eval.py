"""N2: the evaluation oracle vs goldens recorded from the reference's own ThresholdController and
ConcentrationThresholdPredictor (oracle/gen_golden.py eval).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_v20.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def _sd(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("sd/")}


def test_threshold_predictor_matches_reference(gold):
    net = eo.ThresholdPredictorOracle(_sd(gold))
    with torch.no_grad():
        y = net(gold["pred_x"]).numpy()
    assert np.allclose(y, gold["pred_y"], rtol=1e-5, atol=1e-5)


def test_threshold_controller_matches_reference(gold):
    net = eo.ThresholdPredictorOracle(_sd(gold))
    traj, thr, stop_at = gold["traj"], gold["thresholds"], gold["stop_at"]
    assert (stop_at > 0).sum() >= 3 and (stop_at < 0).sum() >= 2          # the fixture exercises both outcomes
    for ep in range(traj.shape[0]):
        ctl = eo.ThresholdControllerOracle(net, gold["scaler_params"])
        ths, stop = [], -1
        for step in range(1, traj.shape[1] + 1):
            if step % 10 == 0:
                ctl.update_threshold(list(traj[ep, :step]))
                ths.append(np.nan if ctl.current_threshold is None else ctl.current_threshold)
            if ctl.should_stop(traj[ep, step - 1], step):
                stop = step
                break
        assert stop == stop_at[ep], (ep, stop, stop_at[ep])
        want = thr[ep][:len(ths)]
        assert np.allclose(np.asarray(ths), want, rtol=1e-5, atol=1e-4, equal_nan=True)


def test_minmax_transform_is_sklearn():
    from sklearn.preprocessing import MinMaxScaler
    rng = np.random.RandomState(0)
    params = rng.rand(5) * 90
    sc = MinMaxScaler().fit(params.reshape(-1, 1))
    v = rng.rand(12) * 120 - 10
    assert np.allclose(eo.minmax_transform(v, params), sc.transform(v.reshape(-1, 1))[:, 0], rtol=0, atol=1e-12)


def test_peak_stop_predictor_is_the_torch_modules():
    """PPOV2.1/evaluate_with_lstm.py:11-27 restated: nn.LSTM(1, 32) -> h_n -> Linear / Linear+Sigmoid."""
    torch.manual_seed(2)
    lstm = torch.nn.LSTM(1, 32, num_layers=1, batch_first=True)
    fc_peak, fc_stop = torch.nn.Linear(32, 1), torch.nn.Sequential(torch.nn.Linear(32, 1), torch.nn.Sigmoid())
    sd = {f"lstm.{k}": v.detach().numpy() for k, v in lstm.state_dict().items()}
    sd.update({"fc_peak.weight": fc_peak.weight.detach().numpy(), "fc_peak.bias": fc_peak.bias.detach().numpy(),
               "fc_stop.0.weight": fc_stop[0].weight.detach().numpy(), "fc_stop.0.bias": fc_stop[0].bias.detach().numpy()})
    x = torch.rand(5, 20, 1)
    with torch.no_grad():
        _, (hn, _) = lstm(x)
        want_peak, want_stop = fc_peak(hn[-1]).squeeze(-1), fc_stop(hn[-1]).squeeze(-1)
        peak, stop = eo.PeakStopPredictorOracle(sd)(x.numpy())
    assert torch.allclose(peak, want_peak, atol=1e-6) and torch.allclose(stop, want_stop, atol=1e-6)
