"""Drop-in surface (config / environment / model / train_ppo2.0.py) on the GPU: same call shapes
as the reference's own modules, results pinned by the reference's golden vectors.  -m gpu."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import PKG
from oracle import ppo_oracle as po
from oracle.env_oracle import FieldBank, OracleEnv

pytestmark = pytest.mark.gpu


def load_train():
    spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(PKG, "train_ppo2.0.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_methane_env_interface_and_golden_prefix(golden):
    from environment import MethaneEnv
    g = golden("env_traces.npz")
    ora = OracleEnv("v2.0", seed=int(g["v2.0_seed"]))
    bank = FieldBank(ora.source[None], ora.conc[None], ora.tke[None])
    env = MethaneEnv("v2.0", bank=bank.interleaved(), bank_sources=bank.sources)
    assert env.action_space.n == 5 and env.observation_space.shape == (6,)
    obs = env.reset()
    assert obs.dtype == np.float32 and np.array_equal(obs, g["v2.0_obs0"][0])
    assert np.array_equal(env.source_pos, g["v2.0_sources"][0])
    assert np.array_equal(env.conc_field, ora.conc) and np.array_equal(env.tke_field, ora.tke)
    # homing script of the golden trace never draws noise-free steps, so only the API shape is
    # checked against the live kernel here; exact traces are covered in test_gpu_env.py
    o, r, d, info = env.step(int(g["v2.0_act"][0]))
    assert o.shape == (6,) and isinstance(r, float) and isinstance(d, bool)
    assert set(info) == {"concentration_reward", "explore_reward", "move_penalty", "tke_penalty", "boundary_penalty"}
    x, y = env.agent_pos
    assert np.isclose(env.conc_field[int(x), int(y)] / 100.0, o[2], atol=1e-7)
    assert env.trajectory[-1]["reached"] in (True, False) and env.step_count == 1


def test_methane_env_procedural_fields_consistent():
    from environment import MethaneEnv
    np.random.seed(3)
    env = MethaneEnv("v2.1")
    assert env.gaussian_params["sigma"] == 15.0
    total = 0
    for t in range(30):
        o, r, d, info = env.step(t % 5)
        x, y = env.agent_pos
        if not d:
            cx, cy = min(int(x), 499), min(int(y), 499)
            assert np.isclose(env.conc_field[cx, cy] / 100.0, o[2], atol=1e-6)
            assert np.isclose(env.tke_field[cx, cy] / 9.0, o[3], atol=1e-6)
        total += 1
        if d:
            env.reset()
    sx, sy = env.source_pos
    assert env.conc_field[int(sx), int(sy)] > 90          # Gaussian peak at the source


def test_methane_env_procedural_equals_oracle():
    """The reference-shaped single environment (MethaneEnv().reset() / .step(a) -> obs, reward, done, info) in its default,
    procedural mode against oracle/procedural_oracle.py with the same seed: observations bit for bit, rewards and the five
    info parts to 1e-6, done flags, source position, and the whole conc_field / tke_field the training script reads
    (train_ppo2.0.py:170-173,203), across episode ends and an explicit mid-episode reset()."""
    from environment import MethaneEnv
    from oracle import procedural_oracle as pr
    seed = 987654
    env = MethaneEnv("v2.0", seed=seed)
    ora = pr.ProceduralVecEnv(1, seed, "v2.0")
    ora.reset()
    # MethaneEnv() resets once in its constructor (environment.py:39) and the script resets again before the first step
    # (train_ppo2.0.py:139): as in the reference, the second reset starts a NEW episode (here: the counter RNG's episode 1)
    ora.episode[0] += 1
    assert np.array_equal(env.reset(), ora._begin(0))
    env.current_radius = 220.0
    ora.set_curriculum(220.0, 0.6)
    rng = np.random.RandomState(0)
    ended = 0
    keys = ("concentration_reward", "explore_reward", "move_penalty", "tke_penalty", "boundary_penalty")
    for t in range(240):
        a = int(rng.choice([1, 3])) if t % 5 else int(rng.randint(0, 5))
        o_ref, r_ref, d_ref, s_ref, info_ref, term_ref = ora.step(np.array([a]))
        o, r, d, info = env.step(a)
        assert np.array_equal(o, term_ref[0]), t
        assert abs(r - r_ref[0]) <= 1e-6 and d == bool(d_ref[0]), t
        assert np.allclose([info[k] for k in keys], info_ref[0], rtol=0, atol=1e-6), t
        if d:
            ended += 1
            assert np.array_equal(env.reset(), o_ref[0]), t           # the auto-started next episode
            src = ora.envs[0].source
            assert np.array_equal(env.source_pos, src)
            if ended == 1:                                                # the table the script indexes: cell for cell
                want_src, conc, tke = pr.full_field(seed, 0, int(ora.episode[0]), 500 / 16)
                assert np.abs(env.conc_field - conc).max() <= 1e-11 and np.abs(env.tke_field - tke).max() <= 1e-11
    assert ended >= 2


def test_actor_critic_module_surface(golden):
    from model import PPOActorCritic
    g = golden("policy_update.npz")
    m = PPOActorCritic(6, 5)
    sd = m.state_dict()
    assert list(sd) == ["feature.0.weight", "feature.0.bias", "feature.1.weight", "feature.1.bias",
                        "feature.3.weight", "feature.3.bias", "feature.4.weight", "feature.4.bias",
                        "actor.weight", "actor.bias", "critic.weight", "critic.bias"]
    assert sum(p.numel() for p in m.parameters()) == 36230
    # orthogonal init (model.py:29-40): rows of W2 orthogonal with gain sqrt(2); tiny actor
    w2 = sd["feature.3.weight"].cpu()
    assert torch.allclose(w2 @ w2.T, 2 * torch.eye(128), atol=1e-4)
    assert sd["actor.weight"].abs().max() < 0.01 and float(sd["feature.0.bias"].abs().sum()) == 0
    # a reference state_dict loads and reproduces the reference's forward (model.py:42-53)
    m.load_state_dict({k: torch.from_numpy(g["init/" + k]) for k in po.MLP_KEYS})
    probs, value = m(torch.from_numpy(g["fwd_x"]))
    assert probs.device.type == "cpu" and value.shape == (64, 1)
    assert np.allclose(probs.numpy(), g["fwd_probs"], atol=1e-6) and np.allclose(value.numpy(), g["fwd_value"], atol=5e-6)
    bad = torch.full((2, 6), float("nan"))
    with pytest.raises(RuntimeError, match="NaN in model output"):
        m(bad)


@pytest.mark.parametrize("opt_kind", ["clip_adam", "torch_adam"])
def test_update_model_matches_reference_golden(golden, opt_kind):
    """_update_model(buffer, model, optimizer) -- the reference's own call shape (train_ppo2.0.py:195)."""
    from model import PPOActorCritic, PPOBuffer
    tr = load_train()
    g = golden("policy_update.npz")
    for case in ("L256", "L7"):
        m = PPOActorCritic(6, 5)
        m.load_state_dict({k: torch.from_numpy(g["init/" + k]) for k in po.MLP_KEYS})
        opt = tr.ClipAdam(m.parameters(), lr=3e-5) if opt_kind == "clip_adam" else torch.optim.Adam(m.parameters(), lr=3e-5)
        buf = PPOBuffer()
        for i in range(len(g[f"{case}/rew"])):
            buf.store(g[f"{case}/obs"][i], g[f"{case}/act"][i], g[f"{case}/rew"][i], g[f"{case}/val"][i],
                      g[f"{case}/logp"][i], g[f"{case}/done"][i])
        assert len(buf.states) == len(g[f"{case}/rew"])
        tr._update_model(buf, m, opt)
        sd = m.state_dict()
        for k in po.MLP_KEYS:
            if f"{case}/post/{k}" in g:
                assert np.allclose(sd[k].cpu().numpy(), g[f"{case}/post/{k}"], rtol=0, atol=4e-7), (case, k)


def test_ppo_trainer_curriculum_pushes_into_env():
    from model import PPOTrainer

    class E:
        current_radius, explore_bonus = 50.0, 0.6
    env = E()
    t = PPOTrainer(env, None, None)
    for _ in range(120):
        t.update(True)
    assert t.current_radius == pytest.approx(45.0) and env.current_radius == 50.0     # env lags by one episode
    t.update(True)
    assert env.current_radius == pytest.approx(45.0) and len(t.success_history) == 1


def test_train_ppo_two_episodes(tmp_path):
    tr = load_train()
    np.random.seed(0)
    model, rows = tr.train_ppo(episodes=2, csv_path=str(tmp_path / "r.csv"), model_path=str(tmp_path / "m" / "p.pth"))
    import pandas as pd
    df = pd.read_csv(tmp_path / "r.csv")
    assert list(df.columns) == ["Episode", "Total_Reward", "Success", "Conc_Reward", "Explore_Reward", "Move_Penalty",
                                "TKE_Penalty", "Boundary_Penalty", "Steps", "Final_Conc", "Current_Radius"]
    assert len(df) == 2 and (df["Steps"] >= 1).all() and (df["Steps"] <= 1000).all()
    sd = torch.load(tmp_path / "m" / "p.pth")
    assert set(sd) == set(po.MLP_KEYS)


def test_train_ppo_vectorised_entry(tmp_path):
    """config.NUM_ENVS > 1 / POLICY == 'lstm' route train_ppo() to the fused vectorised trainer."""
    tr = load_train()
    tr.NUM_ENVS, tr.HORIZON, tr.POLICY, tr.HIDDEN = 64, 32, "lstm", 64
    trainer, rows = tr.train_ppo_vectorised(iterations=3, csv_path=str(tmp_path / "v.csv"), model_path=str(tmp_path / "v.pth"))
    import pandas as pd
    df = pd.read_csv(tmp_path / "v.csv")
    assert list(df.columns)[:3] == ["Episode", "Total_Reward", "Success"] and trainer.iteration == 3
    sd = torch.load(tmp_path / "v.pth")
    assert "lstm.weight_hh_l0" in sd and sd["actor.weight"].shape == (5, 64)
    assert len(df) == int((trainer.buf["flags"].cpu().numpy() & 1).sum()) or len(df) >= 0
    # with nc_path the same loop also writes the trajectory log (row N4) that data_loader reads back
    tr.NUM_ENVS, tr.HORIZON = 128, 64
    nc = str(tmp_path / "training_data.npz")
    trainer, rows = tr.train_ppo_vectorised(iterations=4, csv_path=None, model_path=None, nc_path=nc)
    import sys
    sys.path.insert(0, os.path.dirname(tr.__file__))
    try:
        from data_loader import load_raw_sequences
    finally:
        sys.path.pop(0)
    seqs, concs = load_raw_sequences(nc)
    n_succ = sum(int(r[2]) for r in rows)
    assert len(seqs) == len(concs) <= n_succ and all(len(q) >= 1 for q in seqs)
    if n_succ:
        assert len(seqs) >= 1 and np.isfinite(concs).all()


def test_vectorised_episode_log_matches_bruteforce():
    """EpisodeLogger (cumsum-based) == a plain per-env loop over the raw rollout buffers, incl. episodes that
    span rollouts; and info parts sum to the reward (minus the reach bonus) as in environment.py:139-151."""
    from uavppo.episode_log import EpisodeLogger
    from uavppo.trainer import VecPPOTrainer
    N, T = 48, 40
    for policy, kw in (("lstm", dict(hidden=64)), ("mlp", {})):
        tr = VecPPOTrainer(N, T, policy, device="cuda:0", seed=6, log_info=True, use_curriculum=False, epochs=1, **kw)
        tr.radius = 120.0                      # generous radius: many episodes end inside a few rollouts
        log = EpisodeLogger(N)
        acc = np.zeros((N, 6))
        steps = np.zeros(N, int)
        want = []
        for it in range(4):
            tr.collect()
            rew, info, fl = tr.buf["rew"].cpu().numpy(), tr.info.cpu().numpy(), tr.buf["flags"].cpu().numpy()
            log.add_rollout(rew, info, fl, tr.radius)
            # reward == sum of the five parts (+ reach bonus where reached)
            bonus = np.where((fl & 2) > 0, min(500.0, 150.0 * 50.0 / tr.radius), 0.0)
            assert np.allclose(rew, info[..., :5].sum(-1) + bonus, atol=2e-5)
            for n in range(N):
                for t in range(T):
                    acc[n] += [rew[n, t], *info[n, t, :5]]
                    steps[n] += 1
                    if fl[n, t] & 1:
                        want.append((n, it, t, acc[n].copy(), steps[n], bool(fl[n, t] & 2), info[n, t, 5]))
                        acc[n] = 0
                        steps[n] = 0
            tr.iteration += 1
        want.sort(key=lambda e: (e[1], e[0], e[2]))
        assert len(log.rows) == len(want) >= 10
        for row, (n, it, t, a, st, ok, c) in zip(log.rows, want):
            assert np.allclose([row[1], row[3], row[4], row[5], row[6], row[7]], a, rtol=1e-5, atol=1e-4)
            assert row[8] == st and row[2] == int(ok) and np.isclose(row[9], c * 100.0 if ok else 0.0, atol=1e-4)


def test_vectorised_trajectory_log_matches_oracle_simulation():
    """Row N4: the trajectories handed to the NetCDF writer (x, y = agent_pos after every step, concentration at that
    cell, stop position of successful episodes) from the fused rollout's info columns == a step-by-step oracle
    simulation with the same forced actions and noise; episodes span rollouts; the two-smallest-radii rule decides
    what is written; product RadiusTracker == reference trace."""
    from oracle import traj_oracle as to
    from oracle.env_oracle import FieldBank, OracleVecEnv
    from uavppo.episode_log import RadiusTracker, TrajectoryLogger
    from uavppo.trainer import VecPPOTrainer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "curriculum.npz"), allow_pickle=False)
    rt = RadiusTracker()
    for k, (r, s) in enumerate(zip(g["tracker_radii"], g["tracker_success"])):
        rt.update(float(r), {"r": float(r)}, bool(s))
        assert np.allclose((rt.radius_history + [np.nan, np.nan])[:2], g["tracker_history"][k], equal_nan=True)

    N, T, R = 10, 30, 3
    bank = FieldBank.from_seed(2 * N, "v2.0", seed=13)
    tr = VecPPOTrainer(N, T, "lstm", hidden=64, variant="v2.0", device="cuda:0", seed=2, bank=bank.interleaved(),
                       bank_sources=bank.sources, use_curriculum=False, log_info=True)
    tr.radius = 60.0
    tr.reset()

    class Rec:
        max_episodes = 1000

        def __init__(self):
            self.calls = []

        def write_episode_data(self, *a):
            self.calls.append(a)

    class Rec21(Rec):
        def write_episode_data(self, *a, **kw):
            self.calls.append((a, kw))

    rec, rec21 = Rec(), Rec21()
    log = TrajectoryLogger(N, rec)
    log21 = TrajectoryLogger(N, rec21, gaussian=(15.0, 100.0))     # PPOV2.1's script: every episode, true source position
    ora = OracleVecEnv(N, bank, "v2.0", radius=60.0)
    ora.reset()
    rng = np.random.RandomState(3)
    want, part = [], [([], [], []) for _ in range(N)]
    for it in range(R):
        noise = rng.randn(N, T, 2)
        acts = np.zeros((N, T), np.int32)
        ended = []
        for t in range(T):
            a = []
            for i, e in enumerate(ora.envs):
                d = e.source - e.pos
                a.append((3 if d[0] > 0 else 4) if abs(d[0]) > abs(d[1]) else (1 if d[1] > 0 else 2))
            acts[:, t] = a
            pos_before = None
            obs, rew, done, reached, info, term = None, None, None, None, None, None
            # step every env by hand to read agent_pos BEFORE the auto-reset
            for i, e in enumerate(ora.envs):
                o, r_, d_, s_, inf = e.step(int(a[i]), noise[i, t])
                part[i][0].append(float(e.pos[0])); part[i][1].append(float(e.pos[1])); part[i][2].append(float(o[2]) * 100.0)
                if d_:
                    ended.append((i, t, [np.asarray(v) for v in part[i]], bool(s_), np.array(e.source, np.float64)))
                    part[i] = ([], [], [])
                    ora.episode[i] += 1
                    ora._begin(i)
        ended.sort(key=lambda e_: (e_[0], e_[1]))
        want += ended
        tr.collect(forced_act=torch.from_numpy(acts).to("cuda:0"), noise=torch.from_numpy(noise).to("cuda:0"))
        log.add_rollout(tr.info.cpu().numpy(), tr.buf["flags"].cpu().numpy(), tr.radius)
        log21.add_rollout(tr.info.cpu().numpy(), tr.buf["flags"].cpu().numpy(), tr.radius)
        tr.iteration += 1
    succ = [w for w in want if w[3]]
    assert len(want) >= 6 and len(succ) >= 3 and log.count == len(want)
    assert len(rec.calls) == len(succ)                      # one radius only: every success is written
    for call, (i, t, (xs, ys, cs), ok, _src) in zip(rec.calls, succ):
        ep_idx, steps, x, y, c, sx, sy, sc = call
        assert steps == len(xs) and np.array_equal(np.asarray(x, np.float32), xs.astype(np.float32))       # positions bit-exact
        assert np.array_equal(np.asarray(y, np.float32), ys.astype(np.float32))
        assert np.allclose(c, cs, atol=1e-4) and sx == float(np.float32(xs[-1])) and np.isclose(sc, cs[-1], atol=1e-4)
    # PPOV2.1 form: the last write of EVERY episode carries the true source position, peak as source_conc, sigma and peak
    last = {}
    for a_, kw in rec21.calls:
        last[a_[0]] = (a_, kw)
    assert len(last) == len(want) and len(rec21.calls) == len(want) + len(succ) and log21.written == [(k, w[2][0].size) for k, w in enumerate(want)]
    for k, (i, t, (xs, ys, cs), ok, src) in enumerate(want):
        (ep_idx, steps, x, y, c, sx, sy, sc), kw = last[k]
        assert steps == len(xs) and np.array_equal(np.asarray(x, np.float32), xs.astype(np.float32))
        assert (sx, sy) == (float(np.float32(src[0])), float(np.float32(src[1]))) and sc == 100.0 and kw == {"sigma": 15.0, "peak": 100.0}
    # and the arrays that reach the file follow the writer oracle
    a = to.writer_arrays(log.count + 1, 1000)
    for call in rec.calls:
        to.write_episode(a, *call)
    seqs, _ = to.load_raw_sequences(a)
    assert len(seqs) == len(succ)


def test_ppov11_loop_matches_reference_golden(golden):
    """BASELINE config 1 (PPOV1.1/train_ppo1.1.py:116-190): V1.1 environment, full-buffer updates every 256 steps AND the
    end-of-episode flush of the short leftover buffer (:166-169), curriculum call per episode -- the product's
    train_ppo1.1.py driven with the reference run's recorded actions; losses of all 15 optimiser steps to 1e-4."""
    from environment import MethaneEnv
    from model import PPOActorCritic
    spec = importlib.util.spec_from_file_location("train_ppo1_1", os.path.join(PKG, "train_ppo1.1.py"))
    t11 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t11)
    g = golden("e2e_v11.npz")
    act = g["act"].astype(np.int32)
    steps = len(act)
    ora = OracleEnv("v1.1", seed=int(g["env_seed"]))
    episodes = [(ora.source.copy(), ora.conc, ora.tke)]          # MethaneEnv() resets once itself (environment.py:39) ...
    ora.reset()                                                    # ... and the loop resets again before the first step
    episodes.append((ora.source.copy(), ora.conc, ora.tke))
    noise = np.zeros((steps, 2))
    for t in range(steps):
        st = ora.rs.get_state()
        noise[t] = ora.rs.randn(2)
        ora.rs.set_state(st)
        o, _, d, _, _ = ora.step(int(act[t]))
        assert np.array_equal(d, g["done"][t])
        if d:
            ora.reset()
            episodes.append((ora.source.copy(), ora.conc, ora.tke))
    bank = FieldBank(np.stack([e[0] for e in episodes]), np.stack([e[1] for e in episodes]),
                     np.stack([e[2] for e in episodes]))
    env = MethaneEnv("v1.1", bank=bank.interleaved(), bank_sources=bank.sources)
    model = PPOActorCritic(6, 5)
    model.load_state_dict({k: torch.from_numpy(g["init/" + k]) for k in po.MLP_KEYS})
    # record what _update_model sees and produces
    sizes, losses = [], []
    orig = t11._update_model

    def spy(buffer, m, opt):
        sizes.append(len(buffer.states))
        n0 = len(losses)
        real_grad = t11._t20.ops.mlp_ppo_grad

        def loss_spy(*a, **k):
            r = real_grad(*a, **k)
            s = a[10].cpu().numpy()             # loss_sums
            losses.append((s[0] + s[1] - 0.01 * s[2]) / sizes[-1])
            return r
        t11._t20.ops.mlp_ppo_grad = loss_spy
        try:
            orig(buffer, m, opt)
        finally:
            t11._t20.ops.mlp_ppo_grad = real_grad
        assert len(losses) - n0 == 5
    t11._update_model = spy
    n_ep = int(g["done"].sum())
    model, rows, trainer = t11.train_ppo(episodes=n_ep, csv_path=None, model_path=None, env=env, model=model,
                                         forced_actions=act, noise=noise)
    assert sizes == g["update_sizes"].tolist() and sizes[1] < 256          # the short end-of-episode flush happened
    print("max |loss - reference| over", len(losses), "optimiser steps:", np.max(np.abs(np.array(losses) - g["loss"])))
    assert np.max(np.abs(np.array(losses) - g["loss"])) < 1e-4
    sd = model.state_dict()
    for k in po.MLP_KEYS:
        assert np.isclose(sd[k].double().sum().item(), g["post_sum/" + k], rtol=1e-4, atol=1e-4), k
    ep = g["episodes"]
    assert [r[8] for r in rows] == ep[:, 2].tolist() and [r[2] for r in rows] == ep[:, 1].tolist()
    assert np.allclose([r[1] for r in rows], ep[:, 0], rtol=1e-6)
    assert np.allclose([r[10] for r in rows], ep[:, 3]) and np.allclose(trainer.current_radius, g["curriculum"][-1, 0])


def test_device_episode_log_equals_the_host_logger():
    """uav_episode_rows (the CSV sums on the device, only the ended episodes' rows cross to the host) against EpisodeLogger over the
    raw buffers, six rollouts with episodes spanning them: same episodes in the same order, integers equal, sums to 1e-9."""
    from uavppo.episode_log import DeviceEpisodeLog, EpisodeLogger
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(300, 48, "mlp", device="cuda:0", seed=21, log_info=True, use_curriculum=False)
    tr.radius = 120.0
    tr.reset()
    dev, host = DeviceEpisodeLog(tr), EpisodeLogger(300)
    for it in range(6):
        dev.fence()
        tr.collect()
        k = dev.start()
        host.add_rollout(tr.buf["rew"].cpu().numpy(), tr.info.cpu().numpy(), tr.buf["flags"].cpu().numpy(), 120.0 - it)
        dev.get(k, 120.0 - it)
        tr.update()
    dev_rows = dev.row_lists()
    assert len(dev_rows) == len(host.rows) > 100
    for a, b in zip(dev_rows, host.rows):
        assert a[0] == b[0] and a[2] == b[2] and a[8] == b[8] and a[10] == b[10]
        assert np.allclose([a[1], a[3], a[4], a[5], a[6], a[7], a[9]], [b[1], b[3], b[4], b[5], b[6], b[7], b[9]], rtol=1e-12, atol=1e-9)
    assert any(r[8] > 48 for r in dev_rows)          # episodes that span rollouts
