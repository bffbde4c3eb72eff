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

from config import (BATCH_SIZE, CLIP_EPSILON, DIST_BACKEND, ENTROPY_BETA, ENV_VARIANT, EPISODES, EPOCHS, FIELD_BANK, GAE_MODE, GAMMA,
                    GRID_SIZE, HIDDEN, HORIZON, ITERATIONS, LAMBDA, LEARNING_RATE, LOCAL_RANK, MAX_STEPS, NUM_ENVS, NUM_LAYERS,
                    NUM_MINIBATCHES, POLICY, RANK, SEED, TREND_K, WORLD_SIZE)  # noqa: F401
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


def train_ppo(episodes=None, csv_path="training_results2_0.csv", model_path="model/ppo_successful_models.pth"):
    """Reference-shaped single-environment loop (train_ppo2.0.py:110-261) on the HIP kernels; `episodes` = the reference's 2000
    (train_ppo2.0.py:128) when not given.  With NUM_ENVS > 1, another POLICY or WORLD_SIZE > 1 the vectorised loop runs instead;
    what ends it, in this order: an `episodes` argument given HERE by the caller; else config.EPISODES when set; else
    config.ITERATIONS when set; else the reference's 2000 episodes."""
    if NUM_ENVS > 1 or POLICY != "mlp" or WORLD_SIZE > 1:
        if episodes is None and ITERATIONS is None and EPISODES is None:
            episodes = 2000
        return train_ppo_vectorised(csv_path=csv_path, model_path=model_path, episodes=episodes)
    episodes = 2000 if episodes is None else episodes
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


def _init_distributed():
    """One process per GPU (torchrun, or bench.py's launcher, exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): bring up the
    process group BEFORE the trainer is built.  backend "nccl" == RCCL over xGMI; several ranks sharing one
    GPU (tests, rehearsals) use gloo.  Returns (rank, world, device)."""
    import torch.distributed as dist
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", LOCAL_RANK % ndev)
    torch.cuda.set_device(dev)
    if WORLD_SIZE > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if DIST_BACKEND == "nccl":
            dist.init_process_group("nccl", rank=RANK, world_size=WORLD_SIZE, device_id=dev)
        else:
            dist.init_process_group(DIST_BACKEND, rank=RANK, world_size=WORLD_SIZE)
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    if os.environ.get("UAVPPO_COLLECTIVES") == "abi" and world > 1:
        # the iteration's exchanges on the C ABI's own RCCL communicator (include/uavppo.h K9) instead of torch.distributed,
        # which then only carries the communicator's 128-byte id, the CSV gather and the final barrier
        from uavppo import dist_utils
        if not dist_utils.abi_collectives():
            dist_utils.use_abi_collectives(rank, world, dev)
    return rank, world, dev


def train_ppo_vectorised(iterations=None, csv_path="training_results2_0.csv", model_path="model/ppo_successful_models.pth",
                         nc_path=None, episodes=None, log_every=10):
    """NUM_ENVS environments per GPU with the fused kernels, every BASELINE configuration from config.py alone: LSTM or MLP
    policy, stacked layers, TREND_K observation channels (C5), a materialised FIELD_BANK (C4), WORLD_SIZE ranks (env shards,
    RCCL gradient all-reduce).  One CSV row per finished episode with the reference's 11 columns (uavppo/episode_log.py) in
    (iteration, global env, time) order; the run ends after `iterations` rollouts or once `episodes` episodes have finished
    (the reference trains 2000, train_ppo2.0.py:128), whichever is given (config.ITERATIONS / config.EPISODES otherwise).
    CSV, checkpoint and trajectory log are written by rank 0 only.

    The per-iteration host copies (info / flags / rew) ride the side stream into pinned memory while the update runs
    (RolloutMirror); the loop's only host waits are the curriculum's success bits and that copy."""
    import torch.distributed as dist
    from uavppo import field_bank
    from uavppo.episode_log import DeviceEpisodeLog, EpisodeLogger, RolloutMirror
    from uavppo.trainer import VecPPOTrainer
    iterations = ITERATIONS if iterations is None and episodes is None and EPISODES is None else iterations
    episodes = EPISODES if episodes is None else episodes
    rank, world, dev = _init_distributed()
    bank, bank_src = field_bank.load(FIELD_BANK, ENV_VARIANT, dev)
    tr = VecPPOTrainer(NUM_ENVS, HORIZON, POLICY, hidden=HIDDEN, layers=NUM_LAYERS, variant=ENV_VARIANT, seed=SEED, device=dev,
                       gae_mode=GAE_MODE, num_minibatches=NUM_MINIBATCHES, log_info=True, gamma=GAMMA, lam=LAMBDA,
                       clip=CLIP_EPSILON, ent_beta=ENTROPY_BETA, lr=LEARNING_RATE, epochs=EPOCHS, rank=rank, world_size=world,
                       bank=bank, bank_sources=bank_src, trend_k=TREND_K)
    rows_per_iter = []                  # this rank's rows, iteration by iteration (merged in rank order at the end)
    traj = None
    # The CSV rows are reduced on the device (uav_episode_rows: f64 running sums per env row, one 12-double row per ended episode)
    # and only they cross to the host.  With a trajectory log (nc_path) the raw per-step buffers are needed on the host anyway:
    # then they are mirrored whole and the host-side EpisodeLogger takes the sums.
    use_dev_log = not nc_path
    log = DeviceEpisodeLog(tr) if use_dev_log else EpisodeLogger(NUM_ENVS)
    mirror = log if use_dev_log else RolloutMirror(tr)
    if nc_path and world == 1:       # the reference's trajectory log (train_ppo2.0.py:119-125,216-227,259); single rank only
        from netcdf_writer import NetCDFWriter
        from uavppo.episode_log import TrajectoryLogger
        from config import GAUSSIAN_RADIUS, PEAK_CONCENTRATION
        traj = TrajectoryLogger(NUM_ENVS, NetCDFWriter(nc_path, GRID_SIZE, max_episodes=2000, max_steps=MAX_STEPS),
                                gaussian=(GAUSSIAN_RADIUS, PEAK_CONCENTRATION) if ENV_VARIANT == "v2.1" else None)
    it = 0
    pending = None                   # (pinned slot, curriculum mirror slot) of the rollout whose rows are written one iteration later

    # One rank: a rollout's rows get their episode numbers and the reference's column types as soon as they land and go to
    # the CSV file at once -- the host formats rollout i's ~600 rows while the GPU runs iteration i + 1, instead of 40 k rows
    # after the loop.  Several ranks: blocks are kept and merged in rank order at the end.
    rows = []
    stream = _CsvStream(csv_path, episodes) if (world == 1 and csv_path) else None

    def final_rows(block):
        out = []
        for r in (block.tolist() if use_dev_log else block):
            out.append([len(rows) + len(out) + 1, r[1], int(r[2]), *r[3:8], int(r[8]), *r[9:]])
        return out

    def keep(block):
        if world > 1:
            rows_per_iter.append(block)
            return
        new = final_rows(block)
        rows.extend(new)
        if stream is not None:
            stream.add(new)

    def log_rollout(p):
        slot, cslot, radius = p
        if radius is None:
            radius = tr.rollout_radius(cslot)
        if use_dev_log:
            keep(log.get(slot, radius))                      # landed while the update's kernels were queued / running
            return
        host = mirror.get(slot)
        before = len(log.rows)
        log.add_rollout(host["rew"], host["info"], host["flags"], radius)
        keep(log.rows[before:])
        if traj is not None:
            traj.add_rollout(host["info"], host["flags"], radius)

    # The loop never waits for the device except for copies that landed an iteration ago: the curriculum lives on the device
    # (uavppo/trainer.py, device_curriculum), the per-iteration logs are read from pinned mirrors one iteration late, the episode
    # count that ends the run is the lagged one (the run may collect one or two rollouts more than it needs; rows are cut).
    # With several ranks the episode count that ends the run must be one every rank reads identically: the polled mirror
    # (`episodes_lagged`) depends on when each process's copy happened to land, so a rank could leave the loop an iteration before
    # its peers and strand them in the next all-reduce.  There the count comes from a FIXED mirror slot -- the state before the
    # rollout two iterations back, waited for -- which is replicated device state, hence equal on all ranks at this point.
    slots = []                       # mirror slot of every rollout so far (device curriculum)

    def episodes_for_stop():
        if world == 1 or not tr.device_curriculum:
            return tr.episodes_lagged
        return tr.episodes_before_rollout(slots[-2]) if len(slots) >= 2 else 0

    while (iterations is None or it < iterations) and (episodes is None or episodes_for_stop() < episodes):
        mirror.fence()
        tr.collect()
        slot = mirror.start()
        tr.update()
        tr.update_curriculum()
        tr.poll_param_range()
        tr.iteration += 1
        if pending is not None:
            log_rollout(pending)
        pending = (slot, tr.last_mirror_slot, None) if tr.device_curriculum else (slot, None, tr.rollout_radius())
        if tr.device_curriculum:
            slots.append(tr.last_mirror_slot)
            del slots[:-3]
        it += 1
        if log_every and it % log_every == 0:
            pl, vl, ent = tr.losses()              # (also where NaN probabilities raise, on every rank together)
            if rank == 0:
                print(f"It {it} | episodes {tr.episodes_lagged} | radius {tr.radius_lagged:.1f} | policy {pl:.4f} value {vl:.4f} entropy {ent:.4f}")
    if pending is not None:
        log_rollout(pending)
    tr.losses()
    tr.sync_curriculum()             # (raises if a rank ever ended more episodes in one rollout than its success message holds)
    if traj is not None:
        traj.writer.close()
    if world > 1:                        # rank order == global env order (contiguous shards)
        gathered = [None] * world
        dist.all_gather_object(gathered, rows_per_iter)
        for i in range(it):
            for g in gathered:
                rows.extend(final_rows(g[i]))
    if episodes is not None:
        del rows[episodes:]
    if stream is not None:
        stream.close(rows)
    if rank == 0:
        _save(tr.policy.state_dict(), rows, None if stream is not None else csv_path, model_path)
    if world > 1:
        dist.barrier()
    return tr, rows


def _save(state_dict, rows, csv_path, model_path):
    if model_path:
        os.makedirs(os.path.dirname(model_path) or ".", exist_ok=True)
        torch.save({k: v.cpu() for k, v in state_dict.items()}, model_path)     # train_ppo2.0.py:255-256
    if csv_path:                                                                # :257-258
        _write_csv(rows, csv_path)


def _plain(rows):
    """Exact python ints / floats only (csv writes a numpy scalar's repr, "np.float64(...)"), all finite (pandas writes NaN as "")."""
    import math
    from itertools import chain
    return set(map(type, chain.from_iterable(rows))) <= {int, float} and all(map(math.isfinite, chain.from_iterable(rows)))


class _CsvStream:
    """The CSV of _write_csv, written rollout by rollout (at most `limit` rows); close() falls back to the DataFrame writer over
    the whole file if any row ever held something other than a finite python number."""

    def __init__(self, path, limit=None):
        import csv
        self.path, self.limit, self.n, self.plain = path, limit, 0, True
        self.f = open(path, "w", newline="")
        self.w = csv.writer(self.f, lineterminator="\n")
        self.w.writerow(COLUMNS)

    def add(self, rows):
        if self.limit is not None:
            rows = rows[:max(0, self.limit - self.n)]
        self.n += len(rows)
        self.plain = self.plain and _plain(rows)
        if self.plain:
            self.w.writerows(rows)

    def close(self, all_rows):
        self.f.close()
        if not self.plain:
            pd.DataFrame(all_rows, columns=COLUMNS).to_csv(self.path, index=False)


def _write_csv(rows, csv_path):
    """The file `pd.DataFrame(rows, columns=COLUMNS).to_csv(csv_path, index=False)` writes, byte for byte (shortest-repr floats,
    "\\n" line ends), through the csv module: 3.5 x faster on the 40 k rows of a 60-iteration run at 4096 envs, where the
    DataFrame's float formatting was a third of the script's wall time.  Rows holding anything but finite python numbers
    take the DataFrame path."""
    import csv
    plain = _plain(rows)
    if not plain:
        pd.DataFrame(rows, columns=COLUMNS).to_csv(csv_path, index=False)
        return
    with open(csv_path, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        w.writerow(COLUMNS)
        w.writerows(rows)


if __name__ == "__main__":
    os.environ["KMP_DUPLICATE_LIB_OK"] = "TRUE"
    train_ppo()
