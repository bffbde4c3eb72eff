#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

    python3 -B oracle/gen_golden.py            # writes tests/golden/

The reference (/root/reference, read-only) is imported through oracle/_refload.py; every
number stored below is an output of the reference's own code (numpy 2.2.6 / torch
2.10.0+rocm7.0 CPU -- versions are recorded in each file).  Fixtures hold data only:
seeds, injected inputs and expected outputs.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refload  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
VERS = {"numpy": np.__version__, "torch": torch.__version__}


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


# --------------------------------------------------------------------------- env traces
def scripted_action(t, env, mode, rng):
    """Action scripts that exercise: random walk, 'stay', boundary hugging, homing on the source."""
    if mode == "random":
        return int(rng.randint(0, 5))
    if mode == "stay":
        return 0 if t % 3 else int(rng.randint(0, 5))
    if mode == "edge":          # run along x=0 then along y, bouncing into the walls
        return (2, 4, 1, 4, 3)[(t // 40) % 5]
    # homing: greedy axis move toward the source
    d = env.source_pos - env.agent_pos
    if abs(d[0]) > abs(d[1]):
        return 3 if d[0] > 0 else 4
    return 1 if d[1] > 0 else 2


def gen_env():
    out = {}
    for ver, var, steps in (("PPOV2.0", "v2.0", 700), ("PPOV2.1", "v2.1", 700), ("PPOV1.1", "v1.1", 500)):
        _, envm, _, _ = _refload.load(ver)
        seed = {"v2.0": 11, "v2.1": 12, "v1.1": 13}[var]
        np.random.seed(seed)
        env = envm.MethaneEnv()
        rng = np.random.RandomState(100 + seed)
        modes = ["homing", "random", "edge", "stay", "homing", "homing"]
        # curriculum values per episode; the 3rd makes explore_bonus an np.float64 (f64 path)
        curr = [(50.0, 0.6), (50.0, 0.6), (20.0, np.float64(0.4321)), (8.0, np.float64(0.25)),
                (12.5, 0.1), (50.0, 0.6)]
        ep = 0
        rec = {k: [] for k in ("act", "obs", "rew", "done", "reached", "info", "pos", "ep")}
        sources = [env.source_pos.copy()]
        obs0 = [env._get_obs()]
        for t in range(steps):
            a = scripted_action(t, env, modes[min(ep, len(modes) - 1)], rng)
            # cap episode length of the non-terminating scripts so several resets are covered
            o, r, d, info = env.step(a)
            rec["act"].append(a)
            rec["obs"].append(o)
            rec["rew"].append(r)
            rec["done"].append(d)
            rec["reached"].append(env.trajectory[-1]["reached"])
            rec["info"].append([info["concentration_reward"], info["explore_reward"],
                                info["move_penalty"], info["tke_penalty"], info["boundary_penalty"]])
            rec["pos"].append(env.agent_pos.copy())
            rec["ep"].append(ep)
            force = (not d) and env.step_count >= 120
            if d or force:
                ep += 1
                obs0.append(env.reset())
                sources.append(env.source_pos.copy())
                rad, bon = curr[min(ep, len(curr) - 1)]
                env.current_radius, env.explore_bonus = rad, bon
            rec["done"][-1] = bool(d)
            rec.setdefault("reset_after", []).append(bool(d or force))
        out[f"{var}_seed"] = seed
        out[f"{var}_act"] = np.asarray(rec["act"], np.int8)
        out[f"{var}_obs"] = np.asarray(rec["obs"], np.float32)
        out[f"{var}_rew"] = np.asarray(rec["rew"], np.float64)
        out[f"{var}_done"] = np.asarray(rec["done"], bool)
        out[f"{var}_reached"] = np.asarray(rec["reached"], bool)
        out[f"{var}_info"] = np.asarray(rec["info"], np.float64)
        out[f"{var}_pos"] = np.asarray(rec["pos"], np.float32)
        out[f"{var}_reset_after"] = np.asarray(rec["reset_after"], bool)
        out[f"{var}_sources"] = np.asarray(sources, np.float64)
        out[f"{var}_obs0"] = np.asarray(obs0, np.float32)
        out[f"{var}_curr_radius"] = np.asarray([c[0] for c in curr], np.float64)
        out[f"{var}_curr_bonus"] = np.asarray([float(c[1]) for c in curr], np.float64)
        out[f"{var}_curr_bonus_is_f64"] = np.asarray([isinstance(c[1], np.float64) for c in curr])
        print(var, "episodes", ep, "reached", int(np.sum(rec["reached"])))
    np.savez_compressed(os.path.join(OUT, "env_traces.npz"), versions=str(VERS), **out)


# --------------------------------------------------------------------------- policy + update
class Capture:
    """Records total_loss at each .backward() and the norm returned by clip_grad_norm_."""

    def __init__(self):
        self.loss, self.gnorm = [], []

    def __enter__(self):
        self._bw = torch.Tensor.backward
        self._clip = torch.nn.utils.clip_grad_norm_
        cap = self

        def bw(t, *a, **k):
            cap.loss.append(float(t.detach()))
            return cap._bw(t, *a, **k)

        def clip(params, max_norm, *a, **k):
            n = cap._clip(params, max_norm, *a, **k)
            cap.gnorm.append(float(n))
            return n

        torch.Tensor.backward = bw
        torch.nn.utils.clip_grad_norm_ = clip
        return self

    def __exit__(self, *exc):
        torch.Tensor.backward = self._bw
        torch.nn.utils.clip_grad_norm_ = self._clip


SMALL = ("feature.0.bias", "feature.1.weight", "feature.1.bias", "feature.3.bias",
         "feature.4.weight", "feature.4.bias", "actor.weight", "actor.bias",
         "critic.weight", "critic.bias")


def fill_buffer(buf, L, rng, done_at=()):
    obs = rng.rand(L, 6).astype(np.float32)
    act = rng.randint(0, 5, L)
    rew = rng.randn(L).astype(np.float32) * 2
    val = rng.randn(L).astype(np.float32)
    logp = (np.log(0.2) + 0.05 * rng.randn(L)).astype(np.float32)
    done = np.zeros(L, np.float32)
    for i in done_at:
        done[i] = 1.0
    for i in range(L):
        buf.store(obs[i], act[i], rew[i], val[i], logp[i], done[i])
    return obs, act, rew, val, logp, done


def gen_policy_update():
    _, _, mdl, train = _refload.load("PPOV2.0")
    torch.manual_seed(1234)
    model = mdl.PPOActorCritic(6, 5)
    sd0 = sd_to_np(model.state_dict())
    out = {f"init/{k}": v for k, v in sd0.items()}
    rng = np.random.RandomState(5)
    x = rng.rand(64, 6).astype(np.float32)
    x[:4] = [[0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1], [0.998, 0.002, 1, 1.6, 0.5, 0.2], [0.3, 0.9, 0.0, -0.09, 1, 1]]
    with torch.no_grad():
        probs, value = model(torch.from_numpy(x))
    out.update(fwd_x=x, fwd_probs=probs.numpy(), fwd_value=value.numpy())

    cases = {"L256": (256, (17, 100, 101, 255)), "L7": (7, (2, 6)), "L7b": (7, (0, 3)), "L1": (1, ())}
    for name, (L, done_at) in cases.items():
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd0.items()})
        opt = torch.optim.Adam(model.parameters(), lr=3e-5)
        buf = mdl.PPOBuffer()
        obs, act, rew, val, logp, done = fill_buffer(buf, L, rng, done_at)
        torch.manual_seed(99)                      # randperm: a permutation of a full batch
        with Capture() as cap, contextlib.redirect_stdout(io.StringIO()):
            train._update_model(buf, model, opt)
        post = sd_to_np(model.state_dict())
        out.update({f"{name}/obs": obs, f"{name}/act": act.astype(np.int64), f"{name}/rew": rew,
                    f"{name}/val": val, f"{name}/logp": logp, f"{name}/done": done,
                    f"{name}/loss": np.asarray(cap.loss), f"{name}/gnorm": np.asarray(cap.gnorm)})
        names = [n for n, _ in model.named_parameters()]
        for n, p in zip(names, model.parameters()):
            st = opt.state[p]
            if name == "L256" or n in SMALL:
                out[f"{name}/post/{n}"] = post[n]
            out[f"{name}/post_sum/{n}"] = np.float64(post[n].astype(np.float64).sum())
            out[f"{name}/post_abs/{n}"] = np.float64(np.abs(post[n].astype(np.float64) - sd0[n]).sum())
            if n in SMALL:
                out[f"{name}/m/{n}"] = st["exp_avg"].numpy().copy()
                out[f"{name}/v/{n}"] = st["exp_avg_sq"].numpy().copy()
            out[f"{name}/m_sum/{n}"] = np.float64(st["exp_avg"].double().sum())
            out[f"{name}/v_sum/{n}"] = np.float64(st["exp_avg_sq"].double().sum())
        print(name, "loss", cap.loss, "gnorm", cap.gnorm)
    np.savez_compressed(os.path.join(OUT, "policy_update.npz"), versions=str(VERS), **out)


# --------------------------------------------------------------------------- curriculum
def gen_curriculum():
    _, _, mdl, _ = _refload.load("PPOV2.0")

    class E:
        current_radius, explore_bonus = 50.0, 0.6

    out = {}
    rng = np.random.RandomState(3)
    seqs = {
        "all_success": np.ones(400, bool),
        "all_fail": np.zeros(400, bool),
        "mixed70": rng.rand(1000) < 0.7,
        "mixed20_then_90": np.concatenate([rng.rand(360) < 0.9, rng.rand(300) < 0.1, rng.rand(400) < 0.95]),
        "mid": rng.rand(500) < 0.45,
    }
    for name, seq in seqs.items():
        env = E()
        tr = mdl.PPOTrainer(env, None, None)
        trace = []
        with contextlib.redirect_stdout(io.StringIO()):
            for s in seq:
                tr.update(bool(s))
                trace.append([tr.current_radius, tr.explore_bonus, env.current_radius, env.explore_bonus])
        out[f"{name}/seq"] = seq
        out[f"{name}/trace"] = np.asarray(trace, np.float64)
    # RadiusTracker (train_ppo2.0.py:90-108): the two-smallest-radii history over a synthetic (radius, success) stream
    _, _, _, train = _refload.load("PPOV2.0")
    rt = train.RadiusTracker()
    radii = np.concatenate([np.full(30, 50.0), np.full(25, 42.5), np.full(25, 36.1), np.full(20, 42.5), np.full(30, 30.7)])
    succ = rng.rand(radii.size) < 0.6
    hist = []
    for r, sflag in zip(radii, succ):
        rt.update(float(r), {"r": float(r)}, bool(sflag))
        hist.append((rt.radius_history + [np.nan, np.nan])[:2])
    out["tracker_radii"], out["tracker_success"], out["tracker_history"] = radii, succ, np.asarray(hist, np.float64)
    out["tracker_counts"] = np.asarray([[k, len(v)] for k, v in sorted(rt.success_data.items())], np.float64)
    np.savez_compressed(os.path.join(OUT, "curriculum.npz"), versions=str(VERS), **out)


# --------------------------------------------------------------------------- end to end (N=1)
def gen_e2e(n_updates=24):
    """The reference's training loop body (train_ppo2.0.py:138-198,245) driven for n_updates
    buffer flushes under fixed seeds; every arithmetic step is the reference's own objects."""
    cfg, envm, mdl, train = _refload.load("PPOV2.0")
    env_seed, torch_seed = 21, 22
    np.random.seed(env_seed)
    torch.manual_seed(torch_seed)
    env = envm.MethaneEnv()
    model = mdl.PPOActorCritic(6, 5)
    sd0 = sd_to_np(model.state_dict())
    opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)
    buf = mdl.PPOBuffer()
    trainer = mdl.PPOTrainer(env, model, opt)
    rec = {k: [] for k in ("act", "val", "logp", "rew", "done", "reached", "obs")}
    curr = []
    updates = 0
    with Capture() as cap, contextlib.redirect_stdout(io.StringIO()):
        while updates < n_updates:
            state = env.reset()
            done = False
            while not done and updates < n_updates:
                st = torch.FloatTensor(state).unsqueeze(0)
                with torch.no_grad():
                    probs, value = model(st)
                dist = torch.distributions.Categorical(probs)
                a = dist.sample().item()
                nxt, r, done, info = env.step(a)
                lp = dist.log_prob(torch.tensor(a)).item()
                buf.store(state, a, r, value.item(), lp, done)
                rec["obs"].append(np.asarray(state, np.float32))
                rec["act"].append(a)
                rec["val"].append(value.item())
                rec["logp"].append(lp)
                rec["rew"].append(r)
                rec["done"].append(done)
                rec["reached"].append(env.trajectory[-1]["reached"])
                if len(buf.states) >= cfg.BATCH_SIZE:
                    train._update_model(buf, model, opt)
                    buf.clear()
                    updates += 1
                state = nxt
            if done:
                trainer.update(bool(env.trajectory[-1]["reached"]))
                curr.append([trainer.current_radius, trainer.explore_bonus])
    post = sd_to_np(model.state_dict())
    out = {f"init/{k}": v for k, v in sd0.items()}
    out.update({f"post/{k}": post[k] for k in SMALL})
    out.update({f"post_sum/{k}": np.float64(v.astype(np.float64).sum()) for k, v in post.items()})
    out.update(env_seed=env_seed, torch_seed=torch_seed, n_updates=n_updates,
               act=np.asarray(rec["act"], np.int8), val=np.asarray(rec["val"], np.float32),
               logp=np.asarray(rec["logp"], np.float32), rew=np.asarray(rec["rew"], np.float64),
               done=np.asarray(rec["done"], bool), reached=np.asarray(rec["reached"], bool),
               obs=np.asarray(rec["obs"], np.float32),
               loss=np.asarray(cap.loss), gnorm=np.asarray(cap.gnorm), curriculum=np.asarray(curr))
    print("e2e steps", len(rec["act"]), "episodes", len(curr), "loss[:5]", cap.loss[:5])
    np.savez_compressed(os.path.join(OUT, "e2e_v20.npz"), versions=str(VERS), **out)


def gen_eval():
    """N2: the reference's ThresholdController (PPOV2.0/evaluate_with_lstm.py:10-37) driving its own
    ConcentrationThresholdPredictor (PPOV2.0/model.py:203-240, eval mode) over synthetic concentration trajectories:
    thresholds after every update and the step at which should_stop fires.  The script is loaded from the reference
    tree (its main() is guarded); sklearn's MinMaxScaler is the real one."""
    import importlib.util
    cfg, envm, model, _ = _refload.load("PPOV2.0")
    sys.modules["config"], sys.modules["environment"], sys.modules["model"] = cfg, envm, model
    try:
        spec = importlib.util.spec_from_file_location("eval_ref", f"{_refload.REF_ROOT}/PPOV2.0/evaluate_with_lstm.py")
        ev = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ev)
    finally:
        for m in ("config", "environment", "model"):
            sys.modules.pop(m, None)
    from sklearn.preprocessing import MinMaxScaler
    torch.manual_seed(11)
    net = model.ConcentrationThresholdPredictor(hidden_size=32)      # small fixture; the default 128 is covered on the GPU
    with torch.no_grad():                       # xavier init gives tiny outputs; put the prediction in concentration units
        net.fc[4].bias.fill_(52.0)
        net.fc[4].weight.mul_(8.0)
    net.eval()
    rng = np.random.RandomState(5)
    scaler_params = np.array([0.0, 100.0]) + rng.rand(2)
    scaler = MinMaxScaler()
    scaler.fit(scaler_params.reshape(-1, 1))
    out = {f"sd/{k}": v for k, v in sd_to_np(net.state_dict()).items()}
    out["scaler_params"] = scaler_params
    # raw predictor outputs on random windows
    xw = rng.rand(7, 10, 1).astype(np.float32)
    with torch.no_grad():
        out["pred_x"] = xw
        out["pred_y"] = net(torch.from_numpy(xw), lengths=[10] * 7).numpy()
    trajs, thr, stop_at = [], [], []
    for ep in range(8):
        L = 90
        t = np.arange(L)
        peak = 35.0 + 8.0 * ep
        traj = peak / (1.0 + np.exp(-(t - 30 - 3 * ep) / 6.0)) + rng.rand(L) * 4.0     # rising plume + turbulence
        if ep % 3 == 2:
            traj = traj * 0.3                                                        # never reaches the threshold
        ctl = ev.ThresholdController(net, scaler)
        ths, stop = [], -1
        for step in range(1, L + 1):
            if step % 10 == 0:
                ctl.update_threshold(list(traj[:step]))
                ths.append(np.nan if ctl.current_threshold is None else ctl.current_threshold)
            if ctl.should_stop(traj[step - 1], step):
                stop = step
                break
        trajs.append(traj)
        thr.append(np.pad(np.asarray(ths, np.float64), (0, 9 - len(ths)), constant_values=np.nan))
        stop_at.append(stop)
    out.update(traj=np.asarray(trajs), thresholds=np.asarray(thr), stop_at=np.asarray(stop_at))
    print("eval: stop_at", stop_at, "thresholds[0]", thr[0][:4])
    np.savez_compressed(os.path.join(OUT, "eval_v20.npz"), versions=str(VERS), **out)


def gen_train_lstm():
    """N3: the reference's SequenceDataset (PPOV2.0/train_lstm.py:12-50) on synthetic sequences, and three optimiser steps of
    its ConcentrationThresholdPredictor / SmoothL1Loss(beta=2) / AdamW(3e-4) / clip_grad_norm_(1.0) loop body (:84-92) run in
    eval mode (the reference's dropout takes no external masks), plus torch's own ReduceLROnPlateau trace for the scheduler."""
    import importlib.util
    cfg, envm, model, _ = _refload.load("PPOV2.0")
    sys.modules["config"], sys.modules["model"] = cfg, model
    dl = importlib.util.module_from_spec(importlib.util.spec_from_file_location("data_loader", f"{_refload.REF_ROOT}/PPOV2.0/data_loader.py"))
    _refload._install_third_party_stubs()
    sys.modules["data_loader"] = dl
    dl.__spec__.loader.exec_module(dl)
    try:
        spec = importlib.util.spec_from_file_location("train_lstm_ref", f"{_refload.REF_ROOT}/PPOV2.0/train_lstm.py")
        tl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tl)
    finally:
        for m in ("config", "model", "data_loader"):
            sys.modules.pop(m, None)
    rng = np.random.RandomState(17)
    TS = cfg.TRAINING_SIZE
    seqs, concs = [], []
    for i in range(40):
        L = int(rng.randint(4, 60))
        seqs.append(list(np.cumsum(rng.rand(L) * 3.0) + rng.rand() * 5.0))
        concs.append(float(seqs[-1][-1] * (1.0 + 0.2 * rng.rand())))
    ds = tl.SequenceDataset(seqs, concs, TS)
    X = np.stack([ds[i][0].numpy() for i in range(len(ds))])
    Y = np.stack([ds[i][1].numpy() for i in range(len(ds))])[:, 0]
    out = {"seq_lens": np.asarray([len(s) for s in seqs]), "seq_flat": np.concatenate([np.asarray(s) for s in seqs]),
           "source_concs": np.asarray(concs), "X": X, "Y": Y, "data_min": ds.scaler.data_min_, "data_max": ds.scaler.data_max_,
           "training_size": TS}
    torch.manual_seed(23)
    net = model.ConcentrationThresholdPredictor(input_size=1, hidden_size=32)
    net.eval()
    crit = torch.nn.SmoothL1Loss(beta=2.0)
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
    out.update({f"init/{k}": v for k, v in sd_to_np(net.state_dict()).items()})
    xb, yb = torch.from_numpy(X[:24]), torch.from_numpy(Y[:24])
    losses, gnorms = [], []
    for _ in range(3):
        opt.zero_grad()
        o = net(xb.unsqueeze(-1), lengths=[xb.size(1)] * len(xb))
        loss = crit(o, yb)
        loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)))
        opt.step()
        losses.append(float(loss))
    out.update({f"post/{k}": v for k, v in sd_to_np(net.state_dict()).items()})
    out.update(losses=np.asarray(losses), gnorms=np.asarray(gnorms))
    # the same loop body on RAGGED sequences: forward(x, lengths) packs them (model.py:229-240: pack_padded_sequence with
    # enforce_sorted=False, the output at step lengths[i] - 1 of sequence i goes to the head); zero padding behind each length
    net.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("init/")})
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
    lens = [int(v) for v in rng.randint(1, xb.size(1) + 1, size=len(xb))]
    lens[0], lens[1] = xb.size(1), 1                                   # a full-length and a one-step sequence among them
    xr = xb.clone()
    for i, L in enumerate(lens):
        xr[i, L:] = 0.0
    rl, rg = [], []
    for _ in range(3):
        opt.zero_grad()
        o = net(xr.unsqueeze(-1), lengths=lens)
        loss = crit(o, yb)
        loss.backward()
        rg.append(float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)))
        opt.step()
        rl.append(float(loss))
    out.update({f"ragged_post/{k}": v for k, v in sd_to_np(net.state_dict()).items()})
    out.update(ragged_lengths=np.asarray(lens), ragged_x=xr.numpy(), ragged_losses=np.asarray(rl), ragged_gnorms=np.asarray(rg))
    # scheduler trace (torch itself)
    dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=3e-4)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(dummy, mode="min", factor=0.5, patience=5)
    metrics = np.concatenate([np.linspace(5, 3, 8), np.full(9, 3.0), np.linspace(3.0, 2.99995, 10), np.full(14, 2.9999), [1.0]])
    lrs = []
    for m in metrics:
        sch.step(float(m))
        lrs.append(dummy.param_groups[0]["lr"])
    out.update(sched_metrics=metrics, sched_lrs=np.asarray(lrs))
    print("train_lstm: samples", len(ds), "losses", losses, "gnorms", gnorms, "lrs", sorted(set(lrs)))
    np.savez_compressed(os.path.join(OUT, "train_lstm_v20.npz"), versions=str(VERS), **out)


def gen_train_lstm_v21():
    """N3, V2.1 variant: the reference's TrajectoryDataset (PPOV2.1/train_lstm.py:11-74) on synthetic segments under
    random.seed(5), and three steps of its loop body (:104-118: MSELoss + BCELoss, clip_grad_norm_(1.0), AdamW(1e-3, wd 1e-4))
    on the torch modules its PeakAndStopPredictor is made of (the class is local to train(), :84-101)."""
    import importlib.util
    import random
    cfg, envm, model, _ = _refload.load("PPOV2.1")
    rng = np.random.RandomState(29)
    W = 20
    segs = []
    for e in range(30):                      # 30 episodes, 1-3 sliding-window segments each (same source_pos within one)
        src = rng.rand(2) * 100.0
        L = W + int(rng.randint(0, 3))
        pos = src + (rng.rand(L, 2) - 0.5) * (40.0 if e % 2 else 12.0)
        conc = np.cumsum(rng.rand(L) * 4.0)
        for i in range(0, L - W + 1):
            segs.append({"positions": pos[i:i + W], "concentrations": conc[i:i + W], "source_pos": src, "sigma": 15.0})
    sys.modules["config"], sys.modules["model"] = cfg, model
    try:
        spec = importlib.util.spec_from_file_location("train_lstm_ref21", f"{_refload.REF_ROOT}/PPOV2.1/train_lstm.py")
        tl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tl)
        random.seed(5)
        ds = tl.TrajectoryDataset(segs, stop_radius=10, window_size=W)      # its constructor imports config again
    finally:
        for m in ("config", "model"):
            sys.modules.pop(m, None)
    X = np.stack([np.asarray(f, np.float64) for f in ds.features])[:, :, 0]
    out = {"n_seg": len(segs), "positions": np.stack([s["positions"] for s in segs]),
           "concentrations": np.stack([s["concentrations"] for s in segs]), "source_pos": np.stack([s["source_pos"] for s in segs]),
           "X": X, "labels": np.asarray(ds.labels, np.float64), "window": W}
    torch.manual_seed(31)
    lstm = torch.nn.LSTM(1, 32, num_layers=1, batch_first=True)
    fc_peak, fc_stop = torch.nn.Linear(32, 1), torch.nn.Linear(32, 1)
    params = list(lstm.parameters()) + list(fc_peak.parameters()) + list(fc_stop.parameters())
    sd = lambda: {**{f"lstm.{k}": v.detach().numpy().copy() for k, v in lstm.state_dict().items()},
                  "fc_peak.weight": fc_peak.weight.detach().numpy().copy(), "fc_peak.bias": fc_peak.bias.detach().numpy().copy(),
                  "fc_stop.0.weight": fc_stop.weight.detach().numpy().copy(), "fc_stop.0.bias": fc_stop.bias.detach().numpy().copy()}
    out.update({f"init/{k}": v for k, v in sd().items()})
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    xb = torch.tensor(X[:32], dtype=torch.float32).unsqueeze(-1)
    yb = torch.tensor(np.asarray(ds.labels)[:32], dtype=torch.float32)
    losses, gnorms = [], []
    for _ in range(3):
        opt.zero_grad()
        _, (h_n, _) = lstm(xb)
        h = h_n[-1]
        peak, stop = fc_peak(h).squeeze(-1), torch.sigmoid(fc_stop(h)).squeeze(-1)
        loss = torch.nn.MSELoss()(peak, yb[:, 0]) + torch.nn.BCELoss()(stop, yb[:, 1])
        loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(params, 1.0)))
        opt.step()
        losses.append(float(loss))
    out.update({f"post/{k}": v for k, v in sd().items()})
    out.update(losses=np.asarray(losses), gnorms=np.asarray(gnorms))
    np.savez_compressed(os.path.join(OUT, "train_lstm_v21.npz"), **out)
    print("train_lstm_v21: samples", len(ds), "losses", losses, "gnorms", gnorms, "pos labels", float(np.asarray(ds.labels)[:, 1].sum()))



# --------------------------------------------------------------------------- N4: trajectory log
def gen_nc_schema():
    """PPOV2.1/nc_info.txt:1-46 -- the reference-held schema dump of its training_data.nc (written by its own
    check_nc_info.py) -- parsed into a JSON fixture: dimensions; per variable shape, dtype, attributes, value range."""
    import json
    import re
    dims, variables, cur, section = {}, {}, None, None
    for raw in open("/root/reference/PPOV2.1/nc_info.txt", encoding="utf-8"):
        line = raw.rstrip("\n")
        if not line.strip():
            continue
        if not line.startswith(" "):
            section = "dims" if line.startswith("维度") else ("vars" if line.startswith("变量") else None)
            continue
        if section == "dims":
            k, v = line.strip().split(":")
            dims[k.strip()] = int(v)
        elif section == "vars":
            m = re.match(r"^  (\w+): shape=\(([^)]*)\), dtype=(\w+)$", line)
            if m:
                shape = [int(t) for t in m.group(2).replace(" ", "").split(",") if t]
                cur = variables[m.group(1)] = {"shape": shape, "dtype": m.group(3), "attrs": {}}
                continue
            body = line.strip()
            mm = re.match(r"^min=([-\w.]+), max=([-\w.]+)$", body)
            if mm:
                cur["min"], cur["max"] = float(mm.group(1)), float(mm.group(2))
            else:
                k, v = body.split(":", 1)
                cur["attrs"][k.strip()] = v.strip()
    out = {"source": "PPOV2.1/nc_info.txt:1-46", "dimensions": dims, "variables": variables}
    json.dump(out, open(os.path.join(OUT, "nc_schema.json"), "w"), indent=1, ensure_ascii=False)
    print("nc_schema:", dims, list(variables))


def gen_traj():
    """The reference's own NetCDFWriter (PPOV2.0/netcdf_writer.py:4-114 and PPOV2.1/model.py:351-422) and loaders
    (PPOV2.0/data_loader.py:5-22, PPOV2.1/model.py:68-90) run over oracle/_refload.MemDataset, an in-memory stand-in for
    netCDF4.Dataset (absent from the image).  Recorded: the episodes handed to write_episode_data, every variable's array,
    dtype, fill value and attributes afterwards, and what the loaders return from it."""
    import importlib.util
    import json
    out = {}
    rng = np.random.RandomState(11)
    E, S = 7, 40
    episodes = []
    for ep, steps in ((0, 25), (2, 40), (3, 1), (5, 33), (6, 19)):
        episodes.append((ep, steps, rng.rand(steps) * 499, rng.rand(steps) * 499, rng.rand(steps) * 100,
                         float(rng.rand() * 400 + 50), float(rng.rand() * 400 + 50), 100.0))
    out["episodes_idx"] = np.asarray([[e[0], e[1]] for e in episodes], np.int64)
    for k, (ep, steps, x, y, c, sx, sy, sc) in enumerate(episodes):
        out[f"in{k}/x"], out[f"in{k}/y"], out[f"in{k}/c"] = x, y, c
        out[f"in{k}/src"] = np.asarray([sx, sy, sc])
    meta = {}
    for ver in ("PPOV2.0", "PPOV2.1"):
        cfg, envm, mdl, train = _refload.load(ver)
        W = train.NetCDFWriter
        name = f"mem://{ver}.nc"
        w = W(name, 500, max_episodes=E, max_steps=S)
        for (ep, steps, x, y, c, sx, sy, sc) in episodes:
            if ver == "PPOV2.1":
                w.write_episode_data(ep, steps, x, y, c, sx, sy, sc, 15.0, 100.0)
            else:
                w.write_episode_data(ep, steps, x, y, c, sx, sy, sc)
        w.close()
        ds = _refload._MEM_FILES[name]
        meta[ver] = {"global": {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in ds._gattrs.items()},
                     "dimensions": dict(ds.dimensions), "variables": {}}
        for vn, var in ds.variables.items():
            out[f"{ver}/{vn}"] = var.data.copy()
            meta[ver]["variables"][vn] = {"dims": list(var.dimensions), "dtype": str(var.data.dtype), "attrs": dict(var._attrs),
                                          "explicit_fill": bool(var.explicit_fill),
                                          "fill": (None if not var.explicit_fill else
                                                   ("nan" if isinstance(var.fill_value, np.floating) and np.isnan(var.fill_value)
                                                    else float(var.fill_value)))}
        if ver == "PPOV2.0":
            spec = importlib.util.spec_from_file_location("ref_data_loader", "/root/reference/PPOV2.0/data_loader.py")
            dl = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(dl)
            seqs, concs = dl.load_raw_sequences(name)
            out["v20_raw/lens"] = np.asarray([len(q) for q in seqs], np.int64)
            out["v20_raw/flat"] = np.asarray([v for q in seqs for v in q], np.float64)
            out["v20_raw/source_concs"] = np.asarray(concs, np.float64)
        else:
            with contextlib.redirect_stdout(io.StringIO()):
                segs = mdl.load_trajectory_segments(name, window_size=20)
            out["v21_seg/positions"] = np.stack([q["positions"] for q in segs]).astype(np.float64)
            out["v21_seg/concentrations"] = np.stack([q["concentrations"] for q in segs]).astype(np.float64)
            out["v21_seg/source_pos"] = np.stack([q["source_pos"] for q in segs]).astype(np.float64)
            out["v21_seg/sigma"] = np.asarray([q["sigma"] for q in segs], np.float64)
    out["meta_json"] = np.asarray(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "traj_log.npz"), versions=str(VERS), **out)
    print("traj:", {k: v.shape for k, v in out.items() if k.startswith("v2")})


# --------------------------------------------------------------------------- end to end, PPOV1.1 loop shape (BASELINE C1)
def gen_e2e_v11(extra_steps=300):
    """PPOV1.1/train_ppo1.1.py:116-190 driven by the reference's own objects under fixed seeds: ONE full episode
    (MAX_STEPS = 5000: full-buffer updates every 256 steps, then the end-of-episode flush of the SHORT leftover buffer,
    :166-169, then the curriculum call) plus `extra_steps` of the next episode (one more full-buffer update)."""
    cfg, envm, mdl, train = _refload.load("PPOV1.1")
    env_seed, torch_seed = 31, 32
    np.random.seed(env_seed)
    torch.manual_seed(torch_seed)
    env = envm.MethaneEnv()
    model = mdl.PPOActorCritic(6, 5)
    sd0 = sd_to_np(model.state_dict())
    opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)
    buf = mdl.PPOBuffer()
    trainer = mdl.PPOTrainer(env, model, opt)
    rec = {k: [] for k in ("act", "val", "logp", "rew", "done", "obs")}
    update_sizes, curr, ep_rows = [], [], []
    with Capture() as cap, contextlib.redirect_stdout(io.StringIO()):
        for episode in range(2):
            state = env.reset()
            done = False
            total = 0.0
            nstep = 0
            while not done:
                st = torch.FloatTensor(state).unsqueeze(0)
                with torch.no_grad():
                    probs, value = model(st)
                dist = torch.distributions.Categorical(probs)
                a = dist.sample().item()
                lp = dist.log_prob(torch.tensor(a))
                nxt, r, done, info = env.step(a)
                buf.store(state, a, r, value.item(), lp.item(), done)
                for k, v in (("obs", np.asarray(state, np.float32)), ("act", a), ("val", value.item()), ("logp", lp.item()),
                             ("rew", r), ("done", done)):
                    rec[k].append(v)
                if len(buf.states) >= cfg.BATCH_SIZE:
                    update_sizes.append(len(buf.states))
                    train._update_model(buf, model, opt)
                    buf.clear()
                total += r
                state = nxt
                nstep += 1
                if episode == 1 and nstep >= extra_steps:
                    break
            if not done:
                break
            if len(buf.states) > 0:                      # train_ppo1.1.py:166-169
                update_sizes.append(len(buf.states))
                train._update_model(buf, model, opt)
                buf.clear()
            ep_rows.append([total, float(env.trajectory[-1]["reached"]), env.step_count, trainer.current_radius])
            trainer.update(env.trajectory[-1]["reached"])
            curr.append([trainer.current_radius, trainer.explore_bonus])
    post = sd_to_np(model.state_dict())
    out = {f"init/{k}": v for k, v in sd0.items()}
    out.update({f"post_sum/{k}": np.float64(v.astype(np.float64).sum()) for k, v in post.items()})
    out.update(env_seed=env_seed, torch_seed=torch_seed, max_steps=cfg.MAX_STEPS, update_sizes=np.asarray(update_sizes),
               act=np.asarray(rec["act"], np.int8), val=np.asarray(rec["val"], np.float32),
               logp=np.asarray(rec["logp"], np.float32), rew=np.asarray(rec["rew"], np.float64),
               done=np.asarray(rec["done"], bool), obs=np.asarray(rec["obs"], np.float32),
               loss=np.asarray(cap.loss), gnorm=np.asarray(cap.gnorm), curriculum=np.asarray(curr), episodes=np.asarray(ep_rows))
    print("e2e_v11 steps", len(rec["act"]), "update sizes", update_sizes, "optimiser steps", len(cap.loss))
    np.savez_compressed(os.path.join(OUT, "e2e_v11.npz"), versions=str(VERS), **out)

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["env", "policy", "curriculum", "e2e", "eval", "train_lstm", "train_lstm_v21", "nc_schema", "traj",
                             "e2e_v11"]
    if "env" in which:
        gen_env()
    if "policy" in which:
        gen_policy_update()
    if "curriculum" in which:
        gen_curriculum()
    if "e2e" in which:
        gen_e2e()
    if "eval" in which:
        gen_eval()
    if "train_lstm" in which:
        gen_train_lstm()
    if "train_lstm_v21" in which:
        gen_train_lstm_v21()
    if "nc_schema" in which:
        gen_nc_schema()
    if "traj" in which:
        gen_traj()
    if "e2e_v11" in which:
        gen_e2e_v11()
