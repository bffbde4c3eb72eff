"""Per-episode training log of the vectorised trainer: the 11 CSV columns of the reference
(PPOV2.0/train_ppo2.0.py:129-135, filled at :177-183, :200-206, :230-242), one row per finished episode in
(iteration, env, time) order.  Host-side numpy over the rollout buffers (cumulative sums along T; episodes may
span rollouts, so per-env partial sums are carried)."""
from __future__ import annotations

import numpy as np

COLUMNS = ["Episode", "Total_Reward", "Success", "Conc_Reward", "Explore_Reward", "Move_Penalty", "TKE_Penalty",
           "Boundary_Penalty", "Steps", "Final_Conc", "Current_Radius"]


class EpisodeLogger:
    def __init__(self, num_envs):
        self.carry = np.zeros((num_envs, 6), np.float64)     # running [total, conc, explore, move, tke, boundary]
        self.steps = np.zeros(num_envs, np.int64)
        self.count = 0
        self.rows = []

    def add_rollout(self, rew, info, flags, radius):
        """rew [N,T] f32, info [N,T,>=6] f32 (5 reward parts + obs[2] of the step), flags [N,T] u8.  Returns the number of
        rows added.  Only env rows in which an episode ENDED get a running sum along T; the others add their row total to
        the carry (a rollout ends episodes in a small fraction of its rows: the pass over [N, T, 6] is one reduction)."""
        rew, info, flags = np.asarray(rew), np.asarray(info), np.asarray(flags)
        N, T = rew.shape
        n_idx, t_idx = np.nonzero(flags & 1)                              # ended steps, (env, time) order
        tot = np.empty((N, 6), np.float64)
        tot[:, 0] = rew.sum(axis=1, dtype=np.float64)
        tot[:, 1:] = info[..., :5].sum(axis=1, dtype=np.float64)
        added = 0
        if n_idx.size:
            rows, inv = np.unique(n_idx, return_inverse=True)             # env rows with at least one ended episode
            parts = np.concatenate([rew[rows][..., None].astype(np.float64), info[rows][..., :5].astype(np.float64)], axis=2)
            cs = np.cumsum(parts, axis=1)                                 # [rows, T, 6]
            first = np.ones(n_idx.size, bool)
            first[1:] = n_idx[1:] != n_idx[:-1]
            prev_t = np.where(first, -1, np.roll(t_idx, 1))
            base = np.where(prev_t[:, None] >= 0, cs[inv, np.maximum(prev_t, 0)], 0.0)
            sums = cs[inv, t_idx] - base + np.where(first[:, None], self.carry[n_idx], 0.0)
            steps = (t_idx - prev_t) + np.where(first, self.steps[n_idx], 0)
            success = (flags[n_idx, t_idx] & 2) > 0
            final_conc = np.where(success, info[n_idx, t_idx, 5].astype(np.float64) * 100.0, 0.0)   # :203
            for k in range(n_idx.size):
                self.count += 1
                self.rows.append([self.count, sums[k, 0], int(success[k]), sums[k, 1], sums[k, 2], sums[k, 3], sums[k, 4],
                                  sums[k, 5], int(steps[k]), final_conc[k], radius])
            added = int(n_idx.size)
            last = np.zeros(rows.size, np.int64)
            last[inv] = t_idx                                             # later entries overwrite: last end per env row
            # rows with an ended episode restart their carry behind the last end
            self.carry[rows] = cs[:, -1] - cs[np.arange(rows.size), last]
            self.steps[rows] = T - 1 - last
            keep = np.ones(N, bool)
            keep[rows] = False
            self.carry[keep] += tot[keep]
            self.steps[keep] += T
        else:
            self.carry += tot
            self.steps += T
        return added


class DeviceEpisodeLog:
    """EpisodeLogger's rows with the sums taken ON THE DEVICE (uav_episode_rows): one thread per env row walks the rollout's
    rew / info / flags with f64 running sums and appends a 12-double row per ended episode; only those rows (1.5 MB of pinned
    capacity per rollout instead of the 23 MB of raw buffers at C3) cross to the host, on the process's side stream while the
    update runs.  start() right after collect(), fence() before the next collect(), get(slot, radius) any time later: sorts the
    slot's rows into (env, time) order and appends the reference's 11 columns.  Episodes span rollouts: the running sums of the
    episodes in progress stay in a device block."""
    CAP = 16384

    def __init__(self, trainer, slots=2):
        import torch
        from .trainer import _side_stream
        if trainer.info is None:
            raise ValueError("DeviceEpisodeLog needs a trainer built with log_info=True")
        self.tr = trainer
        d = trainer.device
        self.side = _side_stream(d)
        self.carry = torch.zeros(trainer.N, 8, dtype=torch.float64, device=d)
        self.rows_dev = [torch.zeros(self.CAP, 12, dtype=torch.float64, device=d) for _ in range(slots)]
        self.count_dev = [torch.zeros(1, dtype=torch.int32, device=d) for _ in range(slots)]
        self.rows_host = [torch.zeros(self.CAP, 12, dtype=torch.float64).pin_memory() for _ in range(slots)]
        self.count_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(slots)]
        self.done = [torch.cuda.Event() for _ in range(slots)]
        self.ready = torch.cuda.Event()
        self.pending = [False] * slots
        self.k = 0
        self.rows, self.count = [], 0

    def fence(self):
        import torch
        for k, p in enumerate(self.pending):
            if p:
                torch.cuda.current_stream().wait_event(self.done[k])

    def start(self):
        import torch
        from . import ops
        from .dist_utils import env_shard
        k = self.k
        self.k = (k + 1) % len(self.done)
        if self.pending[k]:
            self.done[k].synchronize()
        tr = self.tr
        self.ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready)
            self.count_dev[k].zero_()
            ops.episode_rows(tr.buf["rew"], tr.info, tr.buf["flags"], self.carry, self.rows_dev[k], self.count_dev[k],
                             env_offset=env_shard(tr.rank, tr.N)[0])
            self.count_host[k].copy_(self.count_dev[k], non_blocking=True)
            self.rows_host[k].copy_(self.rows_dev[k], non_blocking=True)
            self.done[k].record(self.side)
        self.pending[k] = True
        return k

    def get(self, k, radius):
        """Append slot k's episodes (global (env, time) order inside the rollout); returns them as a float64 array [c, 11] in the
        reference's column order (numpy all the way: a Python list per row cost 3.5 ms per C3 rollout)."""
        self.done[k].synchronize()
        self.pending[k] = False
        c = int(self.count_host[k][0])
        if c > self.CAP:
            raise RuntimeError(f"DeviceEpisodeLog: {c} episodes ended in one rollout, capacity {self.CAP}")
        r = self.rows_host[k][:c].numpy()
        r = r[np.lexsort((r[:, 1], r[:, 0]))]
        out = np.empty((c, 11), np.float64)
        out[:, 0] = self.count + 1 + np.arange(c)
        out[:, 1] = r[:, 2]                      # Total_Reward
        out[:, 2] = r[:, 9]                      # Success
        out[:, 3:8] = r[:, 3:8]                  # Conc / Explore / Move / TKE / Boundary
        out[:, 8] = r[:, 8]                      # Steps
        out[:, 9] = r[:, 10]                     # Final_Conc
        out[:, 10] = radius
        self.count += c
        self.rows.append(out)
        return out

    def row_lists(self):
        """All rows so far as Python lists with the reference's types (ints for Episode / Success / Steps)."""
        out = []
        for a in self.rows:
            for r in a.tolist():
                r[0], r[2], r[8] = int(r[0]), int(r[2]), int(r[8])
                out.append(r)
        return out


class RolloutMirror:
    """The per-iteration host copies the reference-shaped training log needs (info [N,T,10], flags, rew: 21 + 0.5 + 2 MB at
    C3) without stalling the GPU: issued on the process's ONE side stream (uavppo/trainer.py: _side_stream) right behind the
    rollout into a ring of pinned host slots, so they cross PCIe while the update's kernels run; the host reads slot k
    after waiting for ITS event only, and the next rollout is fenced behind the copy (which finished milliseconds
    earlier).  Replaces three blocking `.cpu().numpy()` calls per iteration."""

    def __init__(self, trainer, slots=2):
        import torch
        from .trainer import _side_stream
        self.tr = trainer
        self.side = _side_stream(trainer.device)
        self.names = [("info", trainer.info), ("flags", trainer.buf["flags"]), ("rew", trainer.buf["rew"])]
        self.host = [{k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in self.names if v is not None} for _ in range(slots)]
        self.done = [torch.cuda.Event() for _ in range(slots)]
        self.ready = torch.cuda.Event()
        self.pending = [False] * slots
        self.k = 0

    def fence(self):
        """Call before the next collect(): the rollout must not overwrite buffers a copy is still reading."""
        import torch
        for k, p in enumerate(self.pending):
            if p:
                torch.cuda.current_stream().wait_event(self.done[k])

    def start(self):
        """Call right after collect(): queue the copies of this rollout; returns the slot."""
        import torch
        k = self.k
        self.k = (k + 1) % len(self.host)
        if self.pending[k]:
            self.done[k].synchronize()           # the host is a whole ring ahead of its own reads: cannot happen in the loop below
        self.ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready)
            for name, dev in self.names:
                if dev is not None:
                    self.host[k][name].copy_(dev, non_blocking=True)
            self.done[k].record(self.side)
        self.pending[k] = True
        return k

    def get(self, k):
        """numpy views of slot k (valid until the slot is reused, `slots` rollouts later)."""
        self.done[k].synchronize()
        self.pending[k] = False
        return {name: t.numpy() for name, t in self.host[k].items()}


class RadiusTracker:
    """train_ppo2.0.py:90-108: the successful episodes per curriculum radius; `radius_history` keeps the two smallest
    radii seen so far (sorted ascending, the largest dropped)."""

    def __init__(self):
        self.radius_history = []
        self.success_data = {}

    def update(self, current_radius, episode_data, is_success):
        if is_success:
            self.success_data.setdefault(current_radius, []).append(episode_data)
            if current_radius not in self.radius_history:
                self.radius_history.append(current_radius)
                self.radius_history.sort()
                if len(self.radius_history) > 2:
                    del self.radius_history[-1]


class TrajectoryLogger:
    """Per-episode trajectories of the vectorised trainer for the reference's NetCDF log (train_ppo2.0.py:166-175,200-227):
    x, y = agent_pos after every step, conc = conc_field at that cell, and for a successful episode the position /
    concentration it stopped at.  Consumes the fused rollout's info [N,T,10] (columns 5..9 = obs[2], x, y, source x, y) and flags;
    episodes may span rollouts, so per-env partial trajectories are carried.  Episodes are numbered in (iteration, env,
    time) order like EpisodeLogger's rows; an episode is written when it succeeded and its radius is one of the tracker's
    two smallest (the reference's rule at :216-227).

    `gaussian` = (sigma, peak) switches to the PPOV2.1 script's behaviour (PPOV2.1/train_ppo2.0.py:205-232): the
    conditional write carries sigma / peak, and EVERY finished episode is then written (again) with the true source position
    and source_conc = peak -- the file PPOV2.1's load_trajectory_segments / TrajectoryDataset are fed from."""

    def __init__(self, num_envs, writer=None, tracker=None, gaussian=None):
        self.gaussian = gaussian
        self.partial = [([], [], []) for _ in range(num_envs)]
        self.writer, self.tracker = writer, tracker if tracker is not None else RadiusTracker()
        self.count = 0
        self.written = []                # (episode index, steps) of the episodes handed to the writer

    def add_rollout(self, info, flags, radius):
        info, flags = np.asarray(info), np.asarray(flags)
        N, T = flags.shape
        x, y, conc = info[..., 6].astype(np.float64), info[..., 7].astype(np.float64), info[..., 5].astype(np.float64) * 100.0
        n_idx, t_idx = np.nonzero(flags & 1)
        start = np.zeros(N, np.int64)
        for n, t in zip(n_idx, t_idx):
            px, py, pc = self.partial[n]
            xs = np.concatenate([px, x[n, start[n]:t + 1]]) if len(px) else x[n, start[n]:t + 1]
            ys = np.concatenate([py, y[n, start[n]:t + 1]]) if len(py) else y[n, start[n]:t + 1]
            cs = np.concatenate([pc, conc[n, start[n]:t + 1]]) if len(pc) else conc[n, start[n]:t + 1]
            self.partial[n] = ([], [], [])
            start[n] = t + 1
            success = bool(flags[n, t] & 2)
            ep = {"steps": len(xs), "x": xs, "y": ys, "conc": cs, "success": success, "current_radius": radius,
                  "source_x": float(xs[-1]) if success else 0.0, "source_y": float(ys[-1]) if success else 0.0,
                  "source_conc": float(cs[-1]) if success else 0.0}
            self.tracker.update(radius, ep, success)
            room = self.writer is not None and self.count < self.writer.max_episodes
            extra = {} if self.gaussian is None else {"sigma": self.gaussian[0], "peak": self.gaussian[1]}
            if success and radius in self.tracker.radius_history and room:
                self.writer.write_episode_data(self.count, ep["steps"], ep["x"], ep["y"], ep["conc"], ep["source_x"],
                                               ep["source_y"], ep["source_conc"], **extra)
                if self.gaussian is None:
                    self.written.append((self.count, ep["steps"]))
            if self.gaussian is not None and room:
                self.writer.write_episode_data(self.count, ep["steps"], ep["x"], ep["y"], ep["conc"], float(info[n, t, 8]),
                                               float(info[n, t, 9]), self.gaussian[1], **extra)
                self.written.append((self.count, ep["steps"]))
            self.count += 1
        for n in range(N):
            if start[n] < T:
                px, py, pc = self.partial[n]
                self.partial[n] = (np.concatenate([px, x[n, start[n]:]]) if len(px) else x[n, start[n]:].copy(),
                                   np.concatenate([py, y[n, start[n]:]]) if len(py) else y[n, start[n]:].copy(),
                                   np.concatenate([pc, conc[n, start[n]:]]) if len(pc) else conc[n, start[n]:].copy())
