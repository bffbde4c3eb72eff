# train_ppo2.0.py -- the PPOV2.0 trainer on MI355X.
#
# Counterpart of the reference's PPOV2.0/train_ppo2.0.py: same `_update_model(buffer, model,
# optimizer)` (:15-88) and `train_ppo()` (:110-261) entry points and the same 11 CSV columns
# (:129-135), built on environment.MethaneEnv / model.PPOActorCritic.  With config.NUM_ENVS > 1
# (or POLICY == "lstm") `train_ppo()` runs the vectorised trainer (uavppo/trainer.py): fused
# persistent rollout, GAE wave scan and the fused clipped-PPO update for NUM_ENVS envs per GPU.
import os

import numpy as np
import pandas as pd
import torch

from config import (BATCH_SIZE, CLIP_EPSILON, ENTROPY_BETA, ENV_VARIANT, EPOCHS, GAE_MODE, GAMMA, GRID_SIZE, HIDDEN,
                    HORIZON, LAMBDA, LEARNING_RATE, MAX_STEPS, NUM_ENVS, NUM_LAYERS, NUM_MINIBATCHES, POLICY, SEED)  # noqa: F401
from environment import MethaneEnv
from model import PPOActorCritic, PPOBuffer, PPOTrainer
from uavppo import ops

COLUMNS = ["Episode", "Total_Reward", "Success", "Conc_Reward", "Explore_Reward", "Move_Penalty", "TKE_Penalty",
           "Boundary_Penalty", "Steps", "Final_Conc", "Current_Radius"]


class ClipAdam:
    """optimizer for _update_model: global-norm clip (0.5) + Adam fused in one HIP pass over the
    flat parameter buffer (train_ppo2.0.py:87-88,114).  Same constructor shape as torch.optim.Adam."""

    def __init__(self, params, lr=LEARNING_RATE, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5):
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.step_count = 0
        self.exp_avg = self.exp_avg_sq = None
        self.gnorm = None

    def zero_grad(self):
        pass

    def step_flat(self, flat, grad):
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
            self.gnorm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.step_count += 1
        ops.clip_adam(flat, grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.max_norm, self.gnorm)


def _update_model(buffer, model, optimizer):
    """GAE (reference quirks kept) -> normalise -> EPOCHS full-batch clipped-PPO steps, all on the GPU."""
    states, actions, rewards, values, log_probs, dones = buffer.get()
    dev = model.core.device
    L = len(rewards)
    d = lambda t: t.to(dev).contiguous()                        # noqa: E731
    rew, val, done = d(rewards)[None], d(values)[None], d(dones)[None]
    adv = ops.gae(rew, val, done, GAMMA, LAMBDA, GAE_MODE)                     # train_ppo2.0.py:18-32
    adv_n, ret = ops.adv_normalise(adv, val, ops.adv_stats(adv))               # :35-40
    x, act, lp = d(states), d(actions.to(torch.int32)), d(log_probs)
    loss_sums = None
    core = model.core
    fused = (core.in_dim, core.h1, core.h2, core.n_act) == (6, 256, 128, 5)    # the reference's sizes: csrc/mlp_fused.hip
    for _ in range(EPOCHS):                                                    # :43 (one minibatch of the whole buffer)
        if fused:        # forward + loss + backward of :55-86 in one on-chip pass
            loss_sums = torch.zeros(4, dtype=torch.float64, device=dev) if loss_sums is None else loss_sums
            grad = ops.mlp_ppo_grad(core.flat, x, act, lp, adv_n.reshape(-1), ret.reshape(-1), val.reshape(-1), 1.0 / L,
                                    CLIP_EPSILON, ENTROPY_BETA, loss_sums, core.grad)
        else:
            heads = core.heads(x)
            loss_sums, dlogits, dvalue = ops.ppo_loss(heads[:, :5].contiguous(), heads[:, 5].contiguous(), act, lp,
                                                      adv_n.reshape(-1), ret.reshape(-1), val.reshape(-1), 1.0 / L,
                                                      CLIP_EPSILON, ENTROPY_BETA)
            grad = core.backward(torch.cat([dlogits, dvalue[:, None]], 1).contiguous())
        if isinstance(optimizer, ClipAdam):
            optimizer.step_flat(model.core.flat, grad)
        else:                                                                  # a torch optimiser built on model.parameters()
            model.publish_grads()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            optimizer.step()
    if loss_sums is not None and loss_sums[3].item() > 0:
        print("Invalid probs detected!")
        raise RuntimeError("NaN in probs")                                     # :58-62


def train_ppo(episodes=2000, csv_path="training_results2_0.csv", model_path="model/ppo_successful_models.pth"):
    """Reference-shaped single-environment loop (train_ppo2.0.py:110-261) on the HIP kernels."""
    if NUM_ENVS > 1 or POLICY != "mlp":
        return train_ppo_vectorised(csv_path=csv_path, model_path=model_path)
    env = MethaneEnv()
    model = PPOActorCritic(6, 5)
    optimizer = ClipAdam(model.parameters(), lr=LEARNING_RATE)
    buffer = PPOBuffer()
    trainer = PPOTrainer(env, model, optimizer)
    gen = torch.Generator(device=model.core.device).manual_seed(SEED)
    rows = []
    for episode in range(episodes):
        state = env.reset()
        done = False
        ep = dict(total=0.0, steps=0, success=False, source_conc=0.0, conc=0.0, explore=0.0, move=0.0, tke=0.0, bnd=0.0)
        radius_at_start = trainer.current_radius
        while not done:
            st = torch.from_numpy(np.asarray(state, np.float32))[None]
            with torch.no_grad():
                probs, value = model(st)
            action = int(torch.multinomial(probs.to(model.core.device), 1, generator=gen).item())
            next_state, reward, done, info = env.step(action)
            q = probs[0] / probs[0].sum()
            logp = float(torch.log(q.clamp(1.1920929e-07, 1 - 1.1920929e-07))[action])
            ep["total"] += reward
            ep["steps"] += 1
            ep["conc"] += info["concentration_reward"]
            ep["explore"] += info["explore_reward"]
            ep["move"] += info["move_penalty"]
            ep["tke"] += info["tke_penalty"]
            ep["bnd"] += info["boundary_penalty"]
            buffer.store(state, action, reward, value.item(), logp, done)
            if len(buffer.states) >= BATCH_SIZE:
                _update_model(buffer, model, optimizer)
                buffer.clear()
            state = next_state
        if env.trajectory[-1]["reached"]:
            ep["source_conc"] = float(env.trajectory[-1]["conc"]) * 100.0
            ep["success"] = True
        rows.append([episode + 1, ep["total"], int(ep["success"]), ep["conc"], ep["explore"], ep["move"], ep["tke"],
                     ep["bnd"], ep["steps"], ep["source_conc"], radius_at_start])
        trainer.update(ep["success"])
        if (episode + 1) % 50 == 0:
            print(f"Ep {episode + 1} | Reward: {ep['total']:.1f} | Radius: {trainer.current_radius:.1f} | Success: {ep['success']}")
    _save(model.state_dict(), rows, csv_path, model_path)
    return model, rows


def train_ppo_vectorised(iterations=200, csv_path="training_results2_0.csv", model_path="model/ppo_successful_models.pth",
                         nc_path=None):
    """NUM_ENVS environments per GPU with the fused kernels; one CSV row per finished episode with the
    reference's 11 columns (uavppo/episode_log.py), in (iteration, env, time) order."""
    from uavppo.episode_log import EpisodeLogger
    from uavppo.trainer import VecPPOTrainer
    tr = VecPPOTrainer(NUM_ENVS, HORIZON, POLICY, hidden=HIDDEN, layers=NUM_LAYERS, variant=ENV_VARIANT, seed=SEED,
                       gae_mode=GAE_MODE, num_minibatches=NUM_MINIBATCHES, log_info=True, gamma=GAMMA, lam=LAMBDA,
                       clip=CLIP_EPSILON, ent_beta=ENTROPY_BETA, lr=LEARNING_RATE, epochs=EPOCHS)
    log = EpisodeLogger(NUM_ENVS)
    traj = None
    if nc_path:                      # the reference's trajectory log (train_ppo2.0.py:119-125,216-227,259)
        from netcdf_writer import NetCDFWriter
        from uavppo.episode_log import TrajectoryLogger
        from config import GAUSSIAN_RADIUS, PEAK_CONCENTRATION
        traj = TrajectoryLogger(NUM_ENVS, NetCDFWriter(nc_path, GRID_SIZE, max_episodes=2000, max_steps=MAX_STEPS),
                                gaussian=(GAUSSIAN_RADIUS, PEAK_CONCENTRATION) if ENV_VARIANT == "v2.1" else None)
    for it in range(iterations):
        radius = tr.radius
        tr.train_iteration()
        info, flags = tr.info.cpu().numpy(), tr.buf["flags"].cpu().numpy()
        log.add_rollout(tr.buf["rew"].cpu().numpy(), info, flags, radius)
        if traj is not None:
            traj.add_rollout(info, flags, radius)
        pl, vl, ent = tr.losses()
        if (it + 1) % 10 == 0:
            print(f"It {it + 1} | episodes {log.count} | radius {tr.radius:.1f} | policy {pl:.4f} value {vl:.4f} entropy {ent:.4f}")
    if traj is not None:
        traj.writer.close()
    _save(tr.policy.state_dict(), log.rows, csv_path, model_path)
    return tr, log.rows


def _save(state_dict, rows, csv_path, model_path):
    if model_path:
        os.makedirs(os.path.dirname(model_path) or ".", exist_ok=True)
        torch.save({k: v.cpu() for k, v in state_dict.items()}, model_path)     # train_ppo2.0.py:255-256
    if csv_path:
        pd.DataFrame(rows, columns=COLUMNS).to_csv(csv_path, index=False)       # :257-258


if __name__ == "__main__":
    os.environ["KMP_DUPLICATE_LIB_OK"] = "TRUE"
    train_ppo()
