# train_ppo1.1.py -- the PPOV1.1 trainer loop (BASELINE config 1: "1 env, MLP policy, plumbing") on the HIP kernels.
#
# Counterpart of the reference's PPOV1.1/train_ppo1.1.py:94-208.  Same `_update_model` (its :19-92 is line for line the
# PPOV2.0 one, so train_ppo2.0.py's is reused) and the same loop shape, which differs from PPOV2.0's in three places:
#   * the environment is the PPOV1.1 variant: position clip 500 - 1e-6 (environment.py:105), MAX_STEPS = 5000 (config.py:7);
#   * whatever is left in the buffer when an episode ends is flushed through `_update_model` as a SHORT buffer (:166-169),
#     so an update never spans two episodes;
#   * the CSV's Final_Conc is the field value under the agent's last position and Current_Radius the radius AFTER the
#     episode's updates (:171-188); no NetCDF log.
import importlib.util
import os

import numpy as np
import torch

from config import BATCH_SIZE, LEARNING_RATE, SEED
from environment import MethaneEnv
from model import PPOActorCritic, PPOBuffer, PPOTrainer

_spec = importlib.util.spec_from_file_location("train_ppo2_0", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                           "train_ppo2.0.py"))
_t20 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_t20)
_update_model, ClipAdam, COLUMNS, _save = _t20._update_model, _t20.ClipAdam, _t20.COLUMNS, _t20._save


def train_ppo(episodes=2000, csv_path="training_results1_3.csv", model_path="model/ppo_successful_models.pth", env=None,
              model=None, forced_actions=None, noise=None, max_steps_total=None):
    """forced_actions / noise (parity tests): recorded action stream and the env's step normals, consumed in order;
    max_steps_total stops after that many env steps (in the middle of an episode, without its flush)."""
    env = env or MethaneEnv("v1.1")
    model = model or PPOActorCritic(6, 5)
    optimizer = ClipAdam(model.parameters(), lr=LEARNING_RATE)
    buffer = PPOBuffer()
    trainer = PPOTrainer(env, model, optimizer)
    gen = torch.Generator(device=model.core.device).manual_seed(SEED)
    rows, t = [], 0
    for episode in range(episodes):
        state = env.reset()
        done = False
        ep = dict(total=0.0, conc=0.0, explore=0.0, move=0.0, tke=0.0, bnd=0.0)
        while not done:
            st = torch.from_numpy(np.asarray(state, np.float32))[None]
            with torch.no_grad():
                probs, value = model(st)
            if forced_actions is not None:
                action = int(forced_actions[t])
            else:
                action = int(torch.multinomial(probs.to(model.core.device), 1, generator=gen).item())
            next_state, reward, done, info = env.step(action, None if noise is None else noise[t])
            q = probs[0] / probs[0].sum()
            logp = float(torch.log(q.clamp(1.1920929e-07, 1 - 1.1920929e-07))[action])
            buffer.store(state, action, reward, value.item(), logp, done)
            t += 1
            if len(buffer.states) >= BATCH_SIZE:                     # train_ppo1.1.py:152-154
                _update_model(buffer, model, optimizer)
                buffer.clear()
            ep["total"] += reward
            ep["conc"] += info["concentration_reward"]
            ep["explore"] += info["explore_reward"]
            ep["move"] += info["move_penalty"]
            ep["tke"] += info["tke_penalty"]
            ep["bnd"] += info["boundary_penalty"]
            state = next_state
            if max_steps_total is not None and t >= max_steps_total and not done:
                return model, rows, trainer
        if len(buffer.states) > 0:                                   # :166-169 leftover experience
            _update_model(buffer, model, optimizer)
            buffer.clear()
        # :171-173 Final_Conc = conc_field[int(agent_pos)] of the ended episode = 100 x obs[2] of its last step (the
        # kernel has already started the next episode, whose field env.conc_field would now show)
        reached = bool(env.trajectory[-1]["reached"])
        rows.append([episode + 1, ep["total"], int(reached), ep["conc"], ep["explore"], ep["move"], ep["tke"], ep["bnd"],
                     env.step_count, float(env.trajectory[-1]["conc"]) * 100.0, trainer.current_radius])
        trainer.update(reached)
        if (episode + 1) % 50 == 0:
            print(f"Ep {episode + 1} | Reward: {ep['total']:.1f} | Conc: {ep['conc']:.1f} | Steps: {env.step_count}")
    _save(model.state_dict(), rows, csv_path, model_path)
    return model, rows, trainer


if __name__ == "__main__":
    train_ppo()
