# model.py -- PPOActorCritic / PPOBuffer / PPOTrainer with the reference's interface
# (PPOV2.0/model.py:17-53, 75-116, 121-164), computing on the HIP kernels.
#
# PPOActorCritic is a torch.nn.Module whose parameters are VIEWS of one flat device buffer in the
# layout csrc/mlp.hip consumes; state_dict() keys are the reference's (feature.{0,1,3,4}.*,
# actor.*, critic.*), so reference .pth files load and `torch.optim.Adam(model.parameters())`
# works unchanged.  The LSTM policies of BASELINE.json live in uavppo/policy.py.
import numpy as np
import torch
import torch.nn as nn

from config import (DECAY_FACTOR, DEVICE, EXPLORE_BONUS, INITIAL_RADIUS, MIN_RADIUS, RADIUS_DECAY, SUCCESS_THRESHOLD,
                    WINDOW_SIZE)  # noqa: F401
from uavppo import ops
from uavppo.curriculum import Curriculum
from uavppo.policy import MLPActorCritic


class PPOActorCritic(nn.Module):
    def __init__(self, input_size, output_size, device=None):
        super().__init__()
        self.feature = nn.Sequential(nn.Linear(input_size, 256), nn.LayerNorm(256), nn.ReLU(),
                                     nn.Linear(256, 128), nn.LayerNorm(128), nn.ReLU())
        self.actor = nn.Linear(128, output_size)
        self.critic = nn.Linear(128, 1)
        self.core = MLPActorCritic(input_size, output_size, device=device or DEVICE)   # orthogonal init, model.py:29-40
        named = self.core.named_views()
        for name, p in self.named_parameters():
            p.data = named[name]                     # parameters become views of the flat HIP buffer
        self._grads = self.core.named_grads()
        self._nan = torch.zeros(1, dtype=torch.int32, device=self.core.device)

    def to(self, *a, **k):                           # the flat buffer stays on the GPU
        return self

    def forward(self, x):
        """x [B, input] (any device) -> (probs [B, A], value [B, 1]) on x's device."""
        src = x.device
        xg = x.detach().to(self.core.device, torch.float32).contiguous()
        heads = self.core.heads(xg)
        logits = heads[:, :self.core.n_act].contiguous()
        if torch.isnan(logits).any():
            print("NaN in logits! Input:", x)
            raise RuntimeError("NaN in model output")             # model.py:47-49
        _, _, probs, _ = ops.policy_sample(logits, forced_act=torch.zeros(len(xg), dtype=torch.int32, device=xg.device),
                                           want_probs=True, nan_count=self._nan)
        value = heads[:, self.core.n_act:].contiguous()
        return probs.to(src), value.to(src)

    def publish_grads(self):
        """Expose the flat gradient buffer as .grad of every parameter (for torch optimisers /
        clip_grad_norm_ used by a reference-shaped script)."""
        for name, p in self.named_parameters():
            p.grad = self._grads[name]


class PPOBuffer:
    """Python-list rollout buffer, PPOV2.0/model.py:75-116 (the vectorised trainer keeps
    (env, T, feat) device tensors instead -- uavppo/trainer.py)."""

    def __init__(self):
        self.states, self.actions, self.rewards = [], [], []
        self.values, self.log_probs, self.dones = [], [], []

    def clear(self):
        for lst in (self.states, self.actions, self.rewards, self.values, self.log_probs, self.dones):
            lst.clear()

    def store(self, state, action, reward, value, log_prob, done):
        self.states.append(np.array(state, dtype=np.float32))
        self.actions.append(int(action))
        self.rewards.append(float(reward))
        self.values.append(float(value))
        self.log_probs.append(float(log_prob))
        self.dones.append(float(done))

    def get(self):
        return (torch.from_numpy(np.stack(self.states, 0)),
                torch.from_numpy(np.array(self.actions, dtype=np.int64)),
                torch.from_numpy(np.array(self.rewards, dtype=np.float32)),
                torch.from_numpy(np.array(self.values, dtype=np.float32)),
                torch.from_numpy(np.array(self.log_probs, dtype=np.float32)),
                torch.from_numpy(np.array(self.dones, dtype=np.float32)))


class PPOTrainer:
    """Curriculum owner, PPOV2.0/model.py:121-164 (logic in uavppo/curriculum.py)."""

    def __init__(self, env, model, optimizer):
        self.env, self.model, self.optimizer = env, model, optimizer
        self.buffer = PPOBuffer()
        self._c = Curriculum()

    current_radius = property(lambda s: s._c.current_radius, lambda s, v: setattr(s._c, "current_radius", v))
    explore_bonus = property(lambda s: s._c.explore_bonus, lambda s, v: setattr(s._c, "explore_bonus", v))
    success_history = property(lambda s: s._c.success_history)

    def update(self, success):
        before = self._c.current_radius
        full = len(self._c.success_history) + 1 >= WINDOW_SIZE
        self._c.update(success)
        self.env.current_radius, self.env.explore_bonus = self._c.env_radius, self._c.env_bonus   # model.py:132-133
        if full:
            print(f"Curriculum Update: radius -> {self._c.current_radius:.1f}")
        return before
