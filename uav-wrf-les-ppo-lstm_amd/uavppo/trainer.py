"""VecPPOTrainer: the PPOV2.0/2.1 training loop (train_ppo2.0.py:110-261) for N vectorised
environments -- rollout collection, GAE, whole-buffer advantage normalisation, EPOCHS passes
of the clipped-PPO update -- every arithmetic step a HIP kernel behind the C ABI.

One process per GPU: environments are sharded by contiguous global index, parameters are
replicated, and per optimiser step ONE all-reduce (RCCL over xGMI) of the flat gradient buffer
keeps them identical; 3 doubles are all-reduced for the reference's whole-buffer advantage
statistics (train_ppo2.0.py:35-39).
"""
from __future__ import annotations


import numpy as np
import torch

from . import ops
from .dist_utils import (SUCC_CAP, abi_collectives, allreduce_adv_stats, allreduce_grad, allreduce_sum, collectives_on, env_shard, exchange_successes, pack_local_successes,
                         unpack_episode_successes)
from .curriculum import Curriculum
from .policy import LSTMActorCritic, MLPActorCritic

# reference hyper-parameters, PPOV2.0/config.py:12-18 and train_ppo2.0.py:87,114
DEFAULTS = dict(gamma=0.99, lam=0.95, clip=0.2, ent_beta=0.01, lr=3e-5, epochs=5, max_grad_norm=0.5)
# operand ranges of the default (fp16-split) LSTM kernels, include/uavppo.h: |w| < 65504, |x| < 4096, |h0| < 64.  The
# trainer switches to the bf16-split kernels at HALF of each limit; parameters move by at most lr per optimiser step, so a
# weight maximum read one iteration late still has that margin.
RANGE_LIMITS = (0.5 * 65504.0, 0.5 * 4096.0, 0.5 * 64.0)
# the fused MLP kernels run their 256 x 128 products on the same fp16 split: its operand a1 = relu(LayerNorm) is bounded by
# sqrt(255) |g1| + |be1|, so max |param| < 2048 keeps it inside fp16's range (include/uavppo.h, uav_mlp_ppo_grad); half of it:
MLP_RANGE_LIMITS = (0.5 * 2048.0, float("inf"), float("inf"))


_SIDE_STREAMS = {}


def _side_stream(device):
    """ONE side stream per device and process, shared by every trainer built in it.  A trainer used to draw its own
    stream from torch's pool; a SECOND trainer in a process (bench.py's old weak -> strong sequence) then ran its success
    exchange on a second pool stream, and with two rank processes sharing one GPU over gloo every iteration of that second
    trainer stalled for 50-900 ms in update() (tools/two_phase_probe.py: `base` 58-245 ms per C3 iteration against 9.6 ms
    in fresh processes; `samestream` -- the second trainer reusing the first one's stream -- 10.7 ms; `nocurr` and
    `mainstream`, which never touch a side stream, 9.0 / 9.8 ms; dropping the first trainer, gc, empty_cache, kernel
    timers and the torch thread count made no difference).  Streams are a per-process resource: keep exactly one."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _DeviceCurriculumView:
    """What callers read off `trainer.curriculum` when the curriculum lives on the device (each access waits for the device)."""

    def __init__(self, tr):
        self._tr = tr

    @property
    def current_radius(self):
        return self._tr.radius

    @property
    def explore_bonus(self):
        return self._tr.bonus

    @property
    def success_history(self):
        self._tr._sync_curriculum()
        return [None] * self._tr._hist_len          # the window's LENGTH is what callers look at; its bits stay on the device


class VecPPOTrainer:
    def __init__(self, num_envs, horizon, policy="lstm", hidden=128, layers=1, variant="v2.0", device="cuda",
                 seed=1234, gae_mode="reference_exact", num_minibatches=1, bank=None, bank_sources=None,
                 rank=0, world_size=1, use_curriculum=True, trend_k=0, log_info=False, device_curriculum=None, **hp):
        self.hp = dict(DEFAULTS)
        self.hp.update(hp)
        self.N, self.T = int(num_envs), int(horizon)
        self.rank, self.world = int(rank), int(world_size)
        self._coll = self.world > 1 or collectives_on()          # the three exchanges of an iteration are issued
        self.device = torch.device(device)
        self.variant, self.seed = variant, int(seed)
        self.gae_mode = gae_mode
        self.num_minibatches = int(num_minibatches)
        if self.N % self.num_minibatches:
            raise ValueError("num_envs must be divisible by num_minibatches (minibatches are whole env sequences)")
        self.kind = policy
        self.trend_k = int(trend_k)
        self.obs_dim = D = 6 + self.trend_k      # 6 reference features + trend channels (BASELINE C5)
        if policy == "lstm":
            self.policy = LSTMActorCritic(D, hidden, layers, 5, self.device, seed=self.seed)
        elif policy == "mlp":
            self.policy = MLPActorCritic(D, 5, device=self.device, seed=self.seed)
        else:
            raise ValueError(policy)
        N, T, d = self.N, self.T, self.device
        f32 = dict(dtype=torch.float32, device=d)
        self.buf = {"obs": torch.zeros(N, T, D, **f32), "act": torch.zeros(N, T, dtype=torch.int32, device=d),
                    "rew": torch.zeros(N, T, **f32), "val": torch.zeros(N, T, **f32), "logp": torch.zeros(N, T, **f32),
                    "done": torch.zeros(N, T, **f32), "flags": torch.zeros(N, T, dtype=torch.uint8, device=d),
                    "keep": torch.ones(N, T, **f32)}
        # optional per-step reward parts (for the reference's per-episode CSV columns, train_ppo2.0.py:129-135)
        self.info = torch.zeros(N, T, 10, **f32) if log_info else None    # reward parts | obs[2] | agent x, y | source x, y
        self.adv = torch.zeros(N, T, **f32)
        self.adv_n = torch.zeros(N, T, **f32)
        self.ret = torch.zeros(N, T, **f32)
        self.last_val = torch.zeros(N, **f32) if gae_mode == "standard" else None
        self.stats3 = torch.zeros(3, dtype=torch.float64, device=d)
        self.loss_sums = torch.zeros(4, dtype=torch.float64, device=d)
        self.gnorm = torch.zeros(1, **f32)
        self.dhead_bias = torch.zeros(6, **f32)
        self.nan_count = torch.zeros(1, dtype=torch.int32, device=d)
        P = self.policy.num_params()
        self.exp_avg = torch.zeros(P, **f32)
        self.exp_avg_sq = torch.zeros(P, **f32)
        self.opt_step = 0
        self.iteration = 0
        # the reference's MLP (6-256-128, 5 actions) has fused persistent kernels (csrc/mlp_fused.hip); other sizes, trend
        # channels and fused_mlp=False take the layer-by-layer path (uav_mlp_fwd / uav_ppo_loss / uav_mlp_bwd + step-wise rollout)
        self.fused_mlp = policy == "mlp" and D == 6
        self.reuse_rollout_forward = True    # epoch 0 adopts the rollout kernel's stash (same parameters)
        self.use_stepper = True              # h = 256 step-wise rollouts through uav_lstm_stepper_* (A/B switch)
        self.use_fused_tail = True           # ... and heads + sample + env step + store of a step as ONE launch (A/B switch)
        self._rollout_forward_valid = False
        self.record = False          # tests: keep (loss_sums, grad norm) of every optimiser step
        self.record_grads = False    # tests: ... and the (all-reduced, unclipped) flat gradient of every optimiser step + its parameters
        self.log = []
        self.grad_log = []
        # range guard of the fp16-split kernels: device maxima [|param|, |obs|, |h0|], mirrored to pinned host memory
        self.ranges = torch.zeros(3, **f32)
        self._ranges_host = torch.zeros(4, 3, dtype=torch.float32).pin_memory()      # ring: the host may run iterations ahead
        self._ranges_evs = [torch.cuda.Event() for _ in range(4)]
        self._pmax_queue = []        # ring slots whose copy of the Adam kernel's max |param| is on its way, oldest first
        self._pmax_next = 0
        self._buffers_own = False    # True between a collect() and the update() that consumes its buffers
        self.arith = "fp16x3"
        self.range_events = 0        # iterations that ran on the wide-range (bf16-split) kernels
        # force_arith: "fp16x3" | "bf16x6" | "f32_mfma" pins the LSTM kernels' arithmetic (A/B runs, reference gradients);
        # None = the range guard decides.  A handle that uav_create started in another mode (UAV_LSTM_F32_MFMA=1 /
        # UAV_LSTM_BF16X6=1 in the environment: a whole-process A/B) keeps it.
        init_mode = ops.Context.get(self.device).initial_arith
        self.force_arith = None if init_mode == "fp16x3" else init_mode
        self._flat_version = -1
        # environments of this rank: global indices [rank*N, (rank+1)*N)
        self.env_state = torch.zeros(ops.env_state_bytes(N), dtype=torch.uint8, device=d)
        self.cur_obs = torch.zeros(N, D, **f32)
        self.bank = None if bank is None else torch.as_tensor(bank, dtype=torch.float64).to(d).contiguous()
        self.bank_sources = None if bank_sources is None else torch.as_tensor(bank_sources, dtype=torch.float64).to(d).contiguous()
        # T1.  device_curriculum (the default with a curriculum): the 120-episode window, radius and bonus live in a device block
        # (uav_curriculum_*), updated by a one-thread kernel on the side stream from the gathered success messages and read by
        # the env kernels at launch -- the loop has no host synchronisation left.  False: the host-side Curriculum class (the
        # reference's arithmetic in Python; what the N = 1 PPOTrainer of model.py uses), one wait per iteration.
        self.device_curriculum = bool(use_curriculum) if device_curriculum is None else bool(device_curriculum and use_curriculum)
        self._radius, self._bonus = 50.0, 0.6
        self._episodes_done = self._successes_done = 0   # finished episodes of the whole job (all ranks)
        self._hist_len = 0
        self.last_success_bits = None
        if self.device_curriculum:
            self._curr = ops.curriculum_state(self.device, self._radius, self._bonus)
            self._curr_host = torch.zeros(4, self._curr.numel(), dtype=torch.uint8).pin_memory()     # ring of lagged host mirrors
            self._curr_evs = [torch.cuda.Event() for _ in range(4)]
            self._curr_queue, self._curr_next = [], 0
            self.last_mirror_slot = 0
            self._curr_done_ev = torch.cuda.Event()
            self._curr_pending = False
            self.curriculum = _DeviceCurriculumView(self)
        else:
            self._curr = None
            self.curriculum = Curriculum() if use_curriculum else None
        # the curriculum's success bits leave on a side stream right behind the rollout and land in pinned host memory
        # while the update runs: the iteration's one host sync (update_curriculum) then waits for a copy that finished
        # milliseconds ago instead of draining the main stream
        self._side = _side_stream(self.device) if use_curriculum else None
        self._succ_host = (torch.zeros(self.world, 4 + SUCC_CAP + 1, dtype=torch.uint8).pin_memory() if use_curriculum else None)
        self._succ_ev = torch.cuda.Event()
        self._pack_ev = torch.cuda.Event()
        self._roll_ev = torch.cuda.Event()
        self._gather_ev = torch.cuda.Event()
        self._succ_pending = False
        self._succ_exchanged = False
        self._succ_msg = None
        self.side_stream_curriculum = True     # False: pack + copy on the main stream inside update_curriculum (A/B, tools/ab_loop.py)
        if policy == "lstm":
            L, H = layers, hidden
            # recurrent state (h | c) and its snapshot at the start of the current rollout (what BPTT starts from): each pair is
            # ONE allocation, so the snapshot before a rollout is one copy launch instead of two
            self._state = torch.zeros(2, L, N, H, **f32)
            self._state0 = torch.zeros(2, L, N, H, **f32)
            self.h, self.c = self._state[0], self._state[1]
            self.h0, self.c0 = self._state0[0], self._state0[1]
            nb = N // self.num_minibatches
            self.work = {"dgates": ops.lstm_dgates(nb, T, H, self.device), "heads": torch.empty(nb, T, 6, **f32)}   # heads: logits | value, written by the sequence kernels
            for l in range(L):
                self.work[f"stash{l}"] = torch.empty(nb, T, 6 * H, **f32)
                self.work[f"y{l}"] = torch.empty(nb, T, H, **f32)
            # (a dy buffer for paths whose backward does not take dheads is allocated on first use: policy.backward)
        else:
            nb = N // self.num_minibatches
            self.work = {"stash": None}      # layer-by-layer path only (770 floats per sample); allocated on first use
            self._mlp_tmp = {"rew": torch.zeros(N, **f32), "done": torch.zeros(N, **f32),
                             "flags": torch.zeros(N, dtype=torch.uint8, device=d), "stash": None}
        if policy == "lstm":
            self.dhead_bias = self.policy.grad_views["head.bias"]       # the loss kernel's column sums ARE this gradient
        self.dheads = torch.empty((N // self.num_minibatches) * T, 6, **f32)
        self.reset()

    # ------------------------------------------------------------------------------------------
    def env_cfg(self):
        return ops.make_env_cfg(self.variant, self._radius, self._bonus, self.seed, self.bank, self.bank_sources,
                                env_offset=env_shard(self.rank, self.N)[0], n_env_total=self.world * self.N,
                                trend_k=self.trend_k, curriculum=self._curr)

    # ------------------------------------------------------------------------------------------ T1 state
    # radius / bonus / episode counters.  Host curriculum: plain attributes.  Device curriculum: the truth is on the device; the
    # getters below WAIT for it (tests, logging at the end of a run); the training loop itself never calls them -- it reads the
    # lagged mirror (`radius_lagged`, `episodes_lagged`: the state as of the rollout before last, copied to pinned memory on
    # the side stream).
    def _sync_curriculum(self):
        if not self.device_curriculum:
            return
        if self._succ_pending and not self._succ_exchanged and not self._coll:
            self._exchange_successes()         # (with several ranks the exchange is a collective: only update() / collect() issue it)
        if self._curr_pending:
            self._curr_done_ev.synchronize()
        d = ops.curriculum_read(self._curr.cpu().numpy())
        if d["overflow"]:
            raise RuntimeError(f"device curriculum: a rank ended more than {SUCC_CAP} episodes in one rollout (message capacity)")
        self._radius = d["radius"]
        self._bonus = np.float64(d["bonus"]) if d["bonus_is_f64"] else d["bonus"]
        self._episodes_done, self._successes_done, self._hist_len = d["episodes"], d["successes"], d["hist_len"]

    def sync_curriculum(self):
        self._sync_curriculum()

    def _poll_curriculum_mirror(self):
        """Newest landed mirror of the device state (never waits)."""
        latest = None
        while self._curr_queue and self._curr_evs[self._curr_queue[0]].query():
            latest = self._curr_queue.pop(0)
        if latest is not None:
            d = ops.curriculum_read(self._curr_host[latest].numpy())
            if d["overflow"]:          # surfaced one or two rollouts late, never at the end of a long run only
                raise RuntimeError(f"device curriculum: a rank ended more than {SUCC_CAP} episodes in one rollout (message capacity)")
            self._radius = d["radius"]
            self._bonus = np.float64(d["bonus"]) if d["bonus_is_f64"] else d["bonus"]
            self._episodes_done, self._successes_done, self._hist_len = d["episodes"], d["successes"], d["hist_len"]

    def rollout_radius(self, k=None):
        """The radius the LAST collected rollout ran with (device curriculum: from its lagged mirror, slot k = the value of
        `last_mirror_slot` taken right after that rollout's update_curriculum(); waits only for that 64-byte copy)."""
        if not self.device_curriculum:
            return self._rollout_radius
        k = self.last_mirror_slot if k is None else k
        self._curr_evs[k].synchronize()
        return ops.curriculum_read(self._curr_host[k].numpy())["radius"]

    def episodes_before_rollout(self, k):
        """Episodes the whole job had finished BEFORE the rollout whose mirror went to slot k (its `last_mirror_slot`): waits for
        that 64-byte copy only.  The mirror is taken on the device between two curriculum updates, from state every rank holds
        identically -- so this value, unlike the polled `episodes_lagged`, is the same on every rank at the same program point:
        what a multi-rank loop must base its stop decision on (a rank that stops alone leaves the others in the next all-reduce)."""
        if not self.device_curriculum:
            return self._episodes_done
        self._curr_evs[k].synchronize()
        d = ops.curriculum_read(self._curr_host[k].numpy())
        if d["overflow"]:
            raise RuntimeError(f"device curriculum: a rank ended more than {SUCC_CAP} episodes in one rollout (message capacity)")
        return d["episodes"]

    @property
    def radius(self):
        self._sync_curriculum()
        return self._radius

    def _before_host_write(self):
        """A host-side write into the device block must land behind the side stream's pending update of it."""
        if self._curr_pending:
            torch.cuda.current_stream().wait_event(self._curr_done_ev)

    @radius.setter
    def radius(self, v):
        self._radius = float(v)
        if self.device_curriculum:
            self._before_host_write()
            self._curr[:32].view(torch.float64)[0] = self._radius

    @property
    def bonus(self):
        self._sync_curriculum()
        return self._bonus

    @bonus.setter
    def bonus(self, v):
        self._bonus = v
        if self.device_curriculum:
            self._before_host_write()
            self._curr[:32].view(torch.float64)[1] = float(v)
            self._curr[:32].view(torch.float64)[2] = 1.0 if isinstance(v, np.float64) else 0.0

    @property
    def radius_lagged(self):
        if self.device_curriculum:
            self._poll_curriculum_mirror()
        return self._radius

    @property
    def episodes_lagged(self):
        if self.device_curriculum:
            self._poll_curriculum_mirror()
        return self._episodes_done

    @property
    def episodes_done(self):
        self._sync_curriculum()
        return self._episodes_done

    @property
    def successes_done(self):
        self._sync_curriculum()
        return self._successes_done

    # ------------------------------------------------------------------------------------------ range guard
    # The fp16-split kernels need |w| < 65504, |x| < 4096, |h0| < 64 (include/uavppo.h).  What can leave that range, and
    # where it is caught without stalling the loop:
    #   parameters   move by <= lr per optimiser step.  The Adam kernel publishes max |param| after every step
    #                (uav_clip_adam's pmax_out); its copy to pinned host memory is polled, never waited for, at the start of
    #                a later rollout (the host runs one to two iterations ahead of the device), against HALF the limit.  Parameters written by
    #                anything else (initialisation, load_state_dict: torch bumps flat._version) are measured before the
    #                next rollout runs on them -- a sync, on such an iteration only.
    #   observations / recurrent state of the trainer's OWN rollouts are bounded by construction (positions, field values
    #                and counters scaled to O(1); |h| < 1), so they are not measured in the loop.  Buffers filled by a
    #                caller (update() without a preceding collect()) ARE measured, synchronously, before the update.
    def _guarded(self):
        """Every LSTM policy: the h = 64 / 128 sequence kernels AND the h = 256 step kernels are fp16-split by default.
        (At h = 256 the wide-range modes run the generic exact-f32 step path -- slow, but never silently out of range.)
        The fused MLP kernels likewise (their wide-range mode is the exact-f32 MFMA form of the same kernels)."""
        return self.kind == "lstm" or self.fused_mlp

    def _decide(self, maxima):
        ok = all(v == v and v < lim for v, lim in zip(maxima, RANGE_LIMITS if self.kind == "lstm" else MLP_RANGE_LIMITS))
        self.arith = "fp16x3" if ok else "bf16x6"
        self.range_events += (not ok)
        if self.force_arith is not None:
            self.arith = self.force_arith
        return ok

    def _measure_params(self):
        """Synchronous probe of the parameters when something other than the Adam kernel wrote them."""
        if self.policy.flat._version != self._flat_version:
            self._pmax_queue.clear()            # maxima published before the foreign write say nothing about it
            ops.absmax(self.policy.flat, out=self.ranges[0:1])
            self._flat_version = self.policy.flat._version
            self._decide([float(self.ranges[0].item()), 0.0, 0.0])

    def poll_param_range(self):
        """Read the newest of the Adam kernel's max |param| copies that has landed (never waits)."""
        latest = None
        while self._pmax_queue and self._ranges_evs[self._pmax_queue[0]].query():
            latest = self._pmax_queue.pop(0)
        if latest is not None and self._guarded():
            self._decide([float(self._ranges_host[latest, 0]), 0.0, 0.0])

    def check_ranges(self):
        """Kernel arithmetic for the update about to be queued (see above); sets it on the device's handle."""
        if not self._guarded():
            return self.arith
        if self.force_arith is not None:
            self.arith = self.force_arith
        if not self._buffers_own:               # foreign buffers: measure everything now
            self._pmax_queue.clear()
            ops.absmax(self.policy.flat, out=self.ranges[0:1])
            if self.kind == "lstm":
                ops.absmax(self.buf["obs"], out=self.ranges[1:2])
                ops.absmax(self.h0, out=self.ranges[2:3])
            self._flat_version = self.policy.flat._version
            self._decide(self.ranges.tolist())
        else:
            self._measure_params()
        ops.set_lstm_arith(self.arith, self.device)
        return self.arith

    def reset(self):
        ops.env_reset(self.env_state, self.N, self.env_cfg(), self.cur_obs)
        if self.kind == "lstm":
            self.h.zero_()
            self.c.zero_()

    # ------------------------------------------------------------------------------------------ R1
    def collect(self, forced_act=None, noise=None):
        """Fill the (env, T, feat) buffers with one rollout of T steps per env."""
        self._rollout_forward_valid = False
        self._rollout_radius = self._radius    # (host curriculum: what this rollout runs with)
        if self._succ_pending:                 # a previous rollout's flags may still be read by the pack kernel on the side stream
            torch.cuda.current_stream().wait_event(self._pack_ev)
        if self.device_curriculum:
            if self._succ_pending and not self._succ_exchanged:     # collect() twice without an update(): finish the first one's exchange
                self._exchange_successes()
            if self._curr_pending:             # the env kernels read radius / bonus from the device state: a GPU-side wait, no host sync
                torch.cuda.current_stream().wait_event(self._curr_done_ev)
        if self._guarded():
            self.poll_param_range()
            self._measure_params()
            if self.force_arith is not None:
                self.arith = self.force_arith
            ops.set_lstm_arith(self.arith, self.device)
            self._buffers_own = True
        wide = self._guarded() and self.arith != "fp16x3"      # uav_rollout exists in the fp16-split form only
        if self.kind == "lstm" and (self.policy.num_layers != 1 or self.policy.hidden not in (64, 128) or self.trend_k or wide):
            self._state0.copy_(self._state)
            self._collect_stepwise_lstm(forced_act, noise)
        elif self.kind == "lstm":
            self._state0.copy_(self._state)
            reuse = self.reuse_rollout_forward and self.num_minibatches == 1
            ops.rollout_lstm(self.env_state, self.N, self.env_cfg(), self.policy.flat, self.policy.hidden, self.T,
                             self.iteration, self.cur_obs, self.h[0], self.c[0], self.buf, last_val=self.last_val,
                             forced_act=forced_act, noise=noise, nan_count=self.nan_count,
                             stash=self.work["stash0"] if reuse else None, y=self.work["y0"] if reuse else None,
                             info=self.info, heads=self.work["heads"] if reuse else None)
            self._rollout_forward_valid = reuse
        elif self.fused_mlp:
            ops.rollout_mlp(self.env_state, self.N, self.env_cfg(), self.policy.flat, self.T, self.iteration, self.cur_obs,
                            self.buf, last_val=self.last_val, forced_act=forced_act, noise=noise, nan_count=self.nan_count,
                            info=self.info)
        else:
            self._collect_stepwise(forced_act, noise)
        if self.curriculum is not None and self.side_stream_curriculum:
            self._roll_ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._roll_ev)
                self._succ_msg = pack_local_successes(self.buf["flags"])
                self._pack_ev.record(self._side)       # the flags have been read: the next rollout may overwrite them
            self._succ_pending = True
            self._succ_exchanged = False
            if not self._coll:
                self._exchange_successes()

    def _exchange_successes(self):
        """All-gather of the packed success bits + copy to pinned host memory, on the side stream.  With several ranks it is
        issued from update(), BEHIND the advantage-statistics all-reduce in program order: RCCL runs a rank's collectives
        in the order they were issued, and that all-reduce must not queue behind the pack kernel."""
        msgs_main = None
        if abi_collectives() and self._coll:
            # ABI carrier: ONE communicator on raw streams.  RCCL runs a rank's collectives in the order its device reaches them, so
            # the all-gather must not sit on another stream than the all-reduces (two ranks could then reach the two collectives in
            # opposite orders and wait for each other): it goes on the MAIN stream, behind the pack kernel, and the side stream
            # picks the gathered messages up again.  (torch.distributed serialises a group's collectives on its own stream.)
            main = torch.cuda.current_stream()
            main.wait_event(self._pack_ev)
            msgs_main = exchange_successes(self._succ_msg)
            self._gather_ev.record(main)
        with torch.cuda.stream(self._side):
            if msgs_main is not None:
                self._side.wait_event(self._gather_ev)
                msgs = msgs_main
            else:
                msgs = exchange_successes(self._succ_msg)
            if tuple(msgs.shape) != tuple(self._succ_host.shape):       # e.g. world_size > 1 without a process group
                raise RuntimeError(f"success exchange returned {tuple(msgs.shape)}, expected {tuple(self._succ_host.shape)}: "
                                   "is torch.distributed initialised for world_size > 1?")
            if self.device_curriculum:
                # lagged host mirror of the state as it was for THIS rollout, then the update itself: one thread walks the
                # messages rank by rank, episode by episode (model.py:131-164)
                k = self._curr_next
                self._curr_next = (k + 1) % len(self._curr_evs)
                if k in self._curr_queue:                       # the host is a whole ring ahead: drop the oldest mirror
                    self._curr_queue.remove(k)
                self._curr_host[k].copy_(self._curr, non_blocking=True)
                self._curr_evs[k].record(self._side)
                self._curr_queue.append(k)
                self.last_mirror_slot = k
                ops.curriculum_update(self._curr, msgs, SUCC_CAP)
                self._curr_done_ev.record(self._side)
                self._curr_pending = True
            else:
                self._succ_host.copy_(msgs, non_blocking=True)
                self._succ_ev.record(self._side)
        self._succ_exchanged = True

    def _collect_stepwise_lstm(self, forced_act=None, noise=None):
        """Stacked / wide LSTM policies (BASELINE C5: h=256 x2): one cell step per layer + heads GEMM +
        sample + env step per time step.  Same buffers and keep semantics as the fused rollout kernel."""
        b = self.buf
        if not hasattr(self, "_st"):
            f32 = dict(dtype=torch.float32, device=self.device)
            self._st = {"rew": torch.zeros(self.N, **f32), "done": torch.zeros(self.N, **f32),
                        "flags": torch.zeros(self.N, dtype=torch.uint8, device=self.device),
                        "act": torch.zeros(self.N, dtype=torch.int32, device=self.device),
                        "keep": torch.ones(self.N, **f32), "work": {}}
        st = self._st
        cfg = self.env_cfg()
        st["keep"].fill_(1.0)
        # h = 256 on the fp16-split arithmetic, one minibatch: the stepper (uav_lstm_stepper_*) -- weights split once per
        # rollout, state kept in its piece planes, stash / y / heads written at [:, t] of the update's own arrays, so that
        # PPO epoch 0 adopts this forward pass.  Otherwise one uav_lstm_fwd call of T = 1 per layer and step.
        stepper = (self.use_stepper and self.policy.hidden == 256 and self.arith == "fp16x3" and self.num_minibatches == 1
                   and ops.lstm_bwd_caps(self.device, self.obs_dim, 256) != 0)      # 0: the handle is not on the fp16 step path
        # everything after the recurrent layers as one launch (uav_rollout_tail), unless the per-step info rows are wanted
        tail = stepper and self.use_fused_tail and self.info is None
        if stepper:
            self.policy.begin_steps(self.h, self.c)
        for t in range(self.T):
            if tail:
                if t == 0:
                    b["obs"][:, 0] = self.cur_obs                  # later rows are written by the tail of the step before
                top = self.policy.step_layers_at(b["obs"], t, self.work, keep=st["keep"])
                v = self.policy.views
                ops.rollout_tail(self.env_state, cfg, top, t, v["head.weight"], v["head.bias"], self.work["heads"], st["act"],
                                 self.cur_obs, b["obs"], st["keep"], b["act"], b["val"], b["logp"], b["keep"], b["rew"], b["done"],
                                 b["flags"], self.nan_count, seed=self.seed, iteration=self.iteration,
                                 index_offset=env_shard(self.rank, self.N)[0],
                                 forced_act=None if forced_act is None else forced_act[:, t].contiguous(),
                                 noise=None if noise is None else noise[:, t].contiguous())
                continue
            if stepper:
                b["obs"][:, t] = self.cur_obs
                heads = self.policy.step_at(b["obs"], t, self.work, self.work["heads"], keep=st["keep"])
            else:
                heads = self.policy.step(self.cur_obs, self.h, self.c, st["keep"], st["work"])
            fa = None if forced_act is None else forced_act[:, t].contiguous()
            if stepper:        # sample from heads[:, t] in place; action / value / log-prob straight into column t
                act = ops.policy_sample_at(self.work["heads"], t, st["act"], b["act"], b["val"], b["logp"], self.nan_count,
                                           seed=self.seed, iteration=self.iteration, index_offset=env_shard(self.rank, self.N)[0],
                                           forced_act=fa)
            else:
                act, logp, _, _ = ops.policy_sample(heads[:, :5].contiguous(), seed=self.seed, counter=t,
                                                    iteration=self.iteration, index_offset=env_shard(self.rank, self.N)[0],
                                                    forced_act=fa, nan_count=self.nan_count)
                b["obs"][:, t] = self.cur_obs
                b["act"][:, t] = act
                b["val"][:, t] = heads[:, 5]
                b["logp"][:, t] = logp
                b["keep"][:, t] = st["keep"]
            nz = None if noise is None else noise[:, t].contiguous()
            if self.info is not None and "info" not in st:
                st["info"] = torch.zeros(self.N, 5, dtype=torch.float32, device=self.device)
                st["term"] = torch.zeros(self.N, self.obs_dim, dtype=torch.float32, device=self.device)
                st["src"] = torch.zeros(self.N, 2, dtype=torch.float64, device=self.device)
            if self.info is not None:
                ops.env_peek(self.env_state, self.N, source=st["src"])           # source of the episode this step belongs to
            ops.env_step(self.env_state, self.N, cfg, act, self.cur_obs, st["rew"], st["done"], st["flags"], noise=nz,
                         info=st.get("info"), term_obs=st.get("term"))
            if self.info is not None:
                self.info[:, t, :5] = st["info"]
                self.info[:, t, 5] = st["term"][:, 2]
                self.info[:, t, 6:8] = st["term"][:, :2] * 500.0       # step-wise path: position from the observation
                self.info[:, t, 8:10] = st["src"]
            if stepper:        # one launch: keep / rew / done / flags -> column t; keep <- 1 - done
                ops.store_transition(t, st["keep"], st["rew"], st["done"], st["flags"], b["keep"], b["rew"], b["done"], b["flags"])
            else:
                b["rew"][:, t] = st["rew"]
                b["done"][:, t] = st["done"]
                b["flags"][:, t] = st["flags"]
                torch.sub(1.0, st["done"], out=st["keep"])      # the recurrent state restarts where an episode ended
        if stepper:
            for l, sp in enumerate(self.policy.steppers(self.N, self.device)):
                self.h[l].copy_(sp.hn)
                self.c[l].copy_(sp.cn)
            self._rollout_forward_valid = self.reuse_rollout_forward
        # hand the state to the next rollout already masked (its keep[:, 0] is 1), like the fused kernel
        self.h.mul_(st["keep"][None, :, None])
        self.c.mul_(st["keep"][None, :, None])
        if self.last_val is not None:
            hh, cc = self.h.clone(), self.c.clone()
            self.last_val.copy_(self.policy.step(self.cur_obs, hh, cc, None, st["work"])[:, 5])

    def _collect_stepwise(self, forced_act=None, noise=None):
        """MLP policy: one policy-forward + sample + env-step launch group per time step
        (train_ppo2.0.py:157-198 with a batch of N states instead of 1)."""
        b, tmp = self.buf, self._mlp_tmp
        cfg = self.env_cfg()
        for t in range(self.T):
            heads = self.policy.heads(self.cur_obs, stash=tmp["stash"])
            tmp["stash"] = self.policy._stash
            logits = heads[:, :5].contiguous()
            fa = None if forced_act is None else forced_act[:, t].contiguous()
            act, logp, _, _ = ops.policy_sample(logits, seed=self.seed, counter=t, iteration=self.iteration,
                                                index_offset=env_shard(self.rank, self.N)[0], forced_act=fa,
                                                nan_count=self.nan_count)
            b["obs"][:, t] = self.cur_obs
            b["act"][:, t] = act
            b["val"][:, t] = heads[:, 5]
            b["logp"][:, t] = logp
            nz = None if noise is None else noise[:, t].contiguous()
            if self.info is not None and "info" not in tmp:
                tmp["info"] = torch.zeros(self.N, 5, dtype=torch.float32, device=self.device)
                tmp["term"] = torch.zeros(self.N, self.obs_dim, dtype=torch.float32, device=self.device)
                tmp["src"] = torch.zeros(self.N, 2, dtype=torch.float64, device=self.device)
            if self.info is not None:
                ops.env_peek(self.env_state, self.N, source=tmp["src"])           # source of the episode this step belongs to
            ops.env_step(self.env_state, self.N, cfg, act, self.cur_obs, tmp["rew"], tmp["done"], tmp["flags"], noise=nz,
                         info=tmp.get("info"), term_obs=tmp.get("term"))
            if self.info is not None:
                self.info[:, t, :5] = tmp["info"]
                self.info[:, t, 5] = tmp["term"][:, 2]
                self.info[:, t, 6:8] = tmp["term"][:, :2] * 500.0
                self.info[:, t, 8:10] = tmp["src"]
            b["rew"][:, t] = tmp["rew"]
            b["done"][:, t] = tmp["done"]
            b["flags"][:, t] = tmp["flags"]
        if self.last_val is not None:
            self.last_val.copy_(self.policy.heads(self.cur_obs, stash=tmp["stash"])[:, 5])

    # ------------------------------------------------------------------------------------------ G1, G2
    def compute_advantages(self):
        b, hp = self.buf, self.hp
        ops.gae(b["rew"], b["val"], b["done"], hp["gamma"], hp["lam"], self.gae_mode, last_val=self.last_val, out=self.adv)
        ops.adv_stats(self.adv, out=self.stats3)
        allreduce_adv_stats(self.stats3)          # (sum, sumsq, count): whole-buffer statistics over all ranks
        ops.adv_normalise(self.adv, b["val"], self.stats3, self.adv_n, self.ret)

    # ------------------------------------------------------------------------------------------ U1-U3
    def update(self):
        """GAE + EPOCHS x num_minibatches optimiser steps (_update_model, train_ppo2.0.py:15-88)."""
        if self.check_ranges() != "fp16x3":
            self._rollout_forward_valid = False
        self.compute_advantages()
        if self._succ_pending and not self._succ_exchanged:
            self._exchange_successes()
        b, hp = self.buf, self.hp
        N, T, M = self.N, self.T, self.num_minibatches
        nb = N // M
        inv_n = 1.0 / float(nb * T * self.world)
        for _ in range(hp["epochs"]):
            for m in range(M):
                sl = slice(m * nb, (m + 1) * nb)
                args = (b["act"][sl].reshape(-1), b["logp"][sl].reshape(-1), self.adv_n[sl].reshape(-1),
                        self.ret[sl].reshape(-1), b["val"][sl].reshape(-1), inv_n, hp["clip"], hp["ent_beta"],
                        self.loss_sums, self.dheads, self.dhead_bias)
                if self.kind == "lstm":
                    if self._rollout_forward_valid:
                        # first optimiser step after a fused rollout: parameters unchanged since the rollout, whose
                        # kernel already wrote this forward pass (stash, y) and its heads
                        L = self.policy.num_layers
                        self.policy.adopt_forward(b["obs"], b["keep"], self.h0, [self.work[f"stash{l}"] for l in range(L)],
                                                  [self.work[f"y{l}"] for l in range(L)])
                        heads = self.work["heads"]
                        self._rollout_forward_valid = False
                    else:
                        heads = self.policy.heads(b["obs"][sl], b["keep"][sl],
                                                  self.h0[:, sl].contiguous() if M > 1 else self.h0,
                                                  self.c0[:, sl].contiguous() if M > 1 else self.c0, self.work)
                    ops.ppo_loss_heads(heads.view(nb * T, -1), *args)
                elif self.fused_mlp:
                    grad = ops.mlp_ppo_grad(self.policy.flat, b["obs"][sl].reshape(nb * T, self.obs_dim), *args[:8],
                                            self.loss_sums, self.policy.grad)
                else:
                    if self.work["stash"] is None:
                        self.work["stash"] = torch.empty(nb * T * (2 * 256 + 2 * 128 + 2), dtype=torch.float32, device=self.device)
                    heads = self.policy.heads(b["obs"][sl].reshape(nb * T, self.obs_dim), stash=self.work["stash"])
                    ops.ppo_loss_heads(heads, *args)
                if self.kind == "lstm":
                    grad = self.policy.backward(self.dheads, self.work, self.dhead_bias)
                elif not self.fused_mlp:
                    grad = self.policy.backward(self.dheads)
                allreduce_grad(grad)              # RCCL sum over ranks; inv_n already holds 1/global count
                if self.record_grads:            # (gradient, parameters it was taken at)
                    self.grad_log.append((grad.clone(), self.policy.flat.clone()))
                self.opt_step += 1
                ops.clip_adam(self.policy.flat, grad, self.exp_avg, self.exp_avg_sq, self.opt_step, hp["lr"],
                              max_norm=hp["max_grad_norm"], gnorm_out=self.gnorm, pmax_out=self.ranges[0:1])
                if self.record:
                    self.log.append((self.loss_sums.clone(), self.gnorm.clone()))
        if self._guarded():                # max |param| of the last Adam step -> pinned host memory, read at a later poll
            if len(self._pmax_queue) == 4:     # the host is four updates ahead of the device: let the oldest copy land
                self._ranges_evs[self._pmax_queue[0]].synchronize()
                self.poll_param_range()
            k = self._pmax_next
            self._pmax_next = (k + 1) % 4
            self._ranges_host[k].copy_(self.ranges, non_blocking=True)
            self._ranges_evs[k].record()
            self._pmax_queue.append(k)
            self._buffers_own = False
        return self.loss_sums

    # ------------------------------------------------------------------------------------------ T1
    def update_curriculum(self):
        """Feed this iteration's finished episodes to the curriculum (host scalars).  The success bits are
        compacted on the device; only they cross to the host / the other ranks, so every rank feeds the same
        global (env, time)-ordered sequence to its replicated curriculum."""
        if self.curriculum is None:
            return
        if self.device_curriculum:             # no host wait: the messages were (or are now) queued, the device does the rest
            if not self._succ_pending:         # flags not produced by collect(): pack them now
                self._roll_ev.record()
                with torch.cuda.stream(self._side):
                    self._side.wait_event(self._roll_ev)
                    self._succ_msg = pack_local_successes(self.buf["flags"])
                    self._pack_ev.record(self._side)
                self._succ_pending, self._succ_exchanged = True, False
            if not self._succ_exchanged:
                self._exchange_successes()
            self._succ_pending = False
            self._poll_curriculum_mirror()
            return
        if not self._succ_pending:             # flags not produced by collect(): pack and exchange them now
            msgs = exchange_successes(pack_local_successes(self.buf["flags"]))
            if tuple(msgs.shape) != tuple(self._succ_host.shape):
                raise RuntimeError(f"success exchange returned {tuple(msgs.shape)}, expected {tuple(self._succ_host.shape)}: "
                                   "is torch.distributed initialised for world_size > 1?")
            self._succ_host.copy_(msgs, non_blocking=True)
            self._succ_ev.record()
        elif not self._succ_exchanged:         # collect() without an update() in between
            self._exchange_successes()
        self._succ_ev.synchronize()
        self._succ_pending = False
        bits = unpack_episode_successes(self._succ_host.numpy(), self.buf["flags"])
        self._episodes_done += int(bits.size)            # over ALL ranks, in global (env, time) order
        self._successes_done += int(bits.sum())
        self.last_success_bits = bits
        self.curriculum.update_many(bits)
        self._radius, self._bonus = self.curriculum.current_radius, self.curriculum.explore_bonus

    def time_rollouts(self, event_pairs):
        """Measurement hook (bench.py): the next len(event_pairs) calls of train_iteration() record a (start, end) pair of
        torch.cuda.Event around their rollout on the current stream -- the loop that is timed stays train_iteration() itself."""
        self._iter_events = list(event_pairs)[::-1]

    def train_iteration(self):
        ev = self._iter_events.pop() if getattr(self, "_iter_events", None) else None
        if ev is not None:
            ev[0].record()
        self.collect()
        if ev is not None:
            ev[1].record()
        sums = self.update()
        self.update_curriculum()
        self.poll_param_range()
        self.iteration += 1
        return sums

    def losses(self):
        """(policy_loss, value_loss, entropy) of the LAST optimiser step; raises on NaN probabilities
        like the reference (train_ppo2.0.py:58-62)."""
        # the rollout's NaN counter travels with the loss sums, so EVERY rank sees every rank's count and they all
        # raise together (a rank raising alone would leave the others waiting in the next all-reduce)
        t = torch.cat([self.loss_sums, self.nan_count.to(torch.float64)])
        if self._coll:
            allreduce_sum(t)
        s = t.cpu().numpy()
        if s[3] > 0 or s[4] > 0:
            raise RuntimeError("NaN in probs")
        n = (self.N // self.num_minibatches) * self.T * self.world
        return s[0] / n, s[1] / n, s[2] / n
