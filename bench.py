#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec (+ PPO updates/sec) of the PPOV2.0 hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8                      # spawns 8 fresh rank processes itself (RCCL), rank 0 prints the line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of BASELINE.json's headline config (C3): a fused
rollout of 4096 envs x 128 steps per GPU over the procedural Gaussian-plume environment with the
LSTM(h=128) actor-critic, the GAE scan + whole-buffer normalisation, 5 epochs of the clipped-PPO
update (forward, loss, BPTT, weight gradients, grad all-reduce, clip+Adam) and the curriculum
update.

--scaling weak (default): every rank owns the config's env count (4096 at C3); value = all ranks' env-steps /
max-over-ranks time.  Started plainly with --gpus N > 1 (launcher mode) the strong-scaling shape (the config's env count
split over the ranks, SURVEY 8d "strong scaling for C3") is timed afterwards in a SECOND set of fresh rank processes and
reported under "strong_scaling"; rank processes themselves (torchrun) time exactly one shape.
--scaling strong: the split shape is the headline value.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

CONFIGS = {   # BASELINE.json configs (per-GPU env count; SURVEY 8 shorthand)
    "c2": dict(num_envs=256, horizon=64, hidden=64, layers=1, variant="v2.0"),
    "c3": dict(num_envs=4096, horizon=128, hidden=128, layers=1, variant="v2.0"),
    # C4: 8192 envs over 8 GPUs = 1024 per GPU, sigma=15 (PPOV2.1), materialised bank of F=64 fields in HBM
    "c4": dict(num_envs=1024, horizon=128, hidden=128, layers=1, variant="v2.1", bank_fields=64),
    # C5: 32768 envs over 8 GPUs = 4096 per GPU, T=256, LSTM h=256 stacked x2, obs 6 + 2 trend channels
    "c5": dict(num_envs=4096, horizon=256, hidden=256, layers=2, variant="v2.1", trend_k=2),
    # the reference's own policy (MLP 6-256-128, model.py:17-53) at the C3 buffer shape
    "mlp": dict(num_envs=4096, horizon=128, hidden=0, layers=0, variant="v2.0", policy="mlp"),
}
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: exact-f32 MFMA = the f32 vector rate
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA ~2.5 PFLOP/s
NHEADS = 6                       # 5 logits + value
# the reference's MLP policy (6-256-128-6): flops per sample of one forward pass, by the pipe that executes them in the fused
# kernels: the 256 x 128 layer (and, in the update, its two transposed products) as three fp16 piece products per f32 product,
# the two small layers on exact-f32 MFMA
MLP_F32_FWD = 2 * (6 * 256 + 128 * 6)
MLP_H3_FWD = 2 * 256 * 128
# algorithmic bytes per env-step: rollout write 44, GAE 20 (12 read + 8 written), update 44 read per epoch
ALG_ROLLOUT_B, ALG_GAE_B, ALG_UPDATE_B = 44, 20, 44


# ------------------------------------------------------------------------------------------------ rank launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, cmd, extra_env=None, timeout=None, capture_rank0=False, shared_gpu=False):
    """Start `cmd` n times as FRESH child processes (one per rank, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment), wait for all, return (largest exit code, rank 0's stdout or None).  The caller must not have touched
    the GPU: a rank process initialises HIP itself, nothing is re-exec'ed from a process that already did.  If one rank
    fails -- or the set outlives `timeout` seconds -- the others are terminated (by PID) so a dead peer never leaves the
    rest waiting in a collective."""
    import tempfile
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    if shared_gpu:
        # gloo (decided by the caller from its PARSED --backend, so `--backend=gloo` counts too) = the rehearsal of several ranks SHARING one GPU: each process then gets two hardware queues instead of
        # HIP's default four.  With the default, two C5 ranks oversubscribed the GPU's queue slots and every dependent
        # launch waited a scheduling quantum (49.8 s per iteration against 0.46 s: DESIGN.md 6, gpurun_out/b2_c5_*).
        env.setdefault("GPU_MAX_HW_QUEUES", "2")
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_WORLD_SIZE=str(n))
    procs = []
    out0 = tempfile.TemporaryFile() if capture_rank0 else None
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=out0 if (r == 0 and out0) else None))
    t0 = time.time()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = max(rc, abs(code) or 1)
        if live and (rc or (timeout and time.time() - t0 > timeout)):
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    text = None
    if out0:
        out0.seek(0)
        text = out0.read().decode(errors="replace")
        out0.close()
    return rc, text


def last_json_line(text):
    for line in reversed((text or "").strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def launch(args, argv):
    """Plain `python bench.py --gpus N` (N > 1): this process never touches the GPU and becomes the launcher.  The headline
    phase runs in one set of fresh rank processes; the strong-scaling shape, when the headline is weak, in a
    SECOND set of fresh processes -- so neither phase can slow down, hang or lose the other (round 2's in-process
    weak -> strong sequence stalled in a shared-GPU rehearsal: DESIGN.md 6).  One JSON line on stdout."""
    me = [sys.executable, os.path.abspath(__file__)]
    t0 = time.time()
    shared = args.backend == "gloo"
    # generous, config-dependent limit: a rank stuck in a collective or on a GPU wait must not hold the launcher (and the
    # one JSON line) forever.  On expiry spawn_ranks terminates the children by PID (never re-executes them), rc = 124.
    per_iter_s = {"c5": 0.25, "c4": 0.02, "c3": 0.02, "c2": 0.01, "mlp": 0.02}.get(args.config, 0.05) * (args.gpus if shared else 1)
    limit = args.headline_timeout or (600.0 + 20.0 * per_iter_s * (args.steps + args.warmup))
    rc, text = spawn_ranks(args.gpus, me + argv, capture_rank0=True, timeout=limit, shared_gpu=shared)
    out = last_json_line(text)
    if rc or out is None:
        why = f"timed out after {limit:.0f} s; rank processes terminated" if rc == 124 else f"rc {rc}"
        sys.stderr.write(f"[bench] headline phase failed ({why})\n")
        return rc or 1
    if args.scaling == "weak" and not args.no_strong_phase and CONFIGS[args.config]["num_envs"] % args.gpus == 0:
        wall = time.time() - t0
        k2, w2 = max(3, args.steps // 2), min(args.warmup, 2)
        argv2 = ["--gpus", str(args.gpus), "--config", args.config, "--backend", args.backend, "--collectives", args.collectives,
                 "--pg-timeout", str(args.pg_timeout), "--scaling", "strong",
                 "--steps", str(k2), "--warmup", str(w2), "--no-cpu-baseline"]
        rc2, text2 = spawn_ranks(args.gpus, me + argv2, capture_rank0=True, timeout=max(180.0, 4.0 * wall), shared_gpu=shared)
        o2 = last_json_line(text2)
        if rc2 == 0 and o2:
            out["strong_scaling"] = {"value": o2["value"], "unit": o2["unit"], "num_envs_per_gpu": o2["config"]["num_envs_per_gpu"],
                                     "num_envs_total": o2["config"]["num_envs_total"], "steps": o2["steps"],
                                     "ms_per_step": o2["ms_per_step"], "rollout_ms": o2["rollout_ms"],
                                     "ranks": "a second set of fresh rank processes"}
        else:
            out["strong_scaling"] = {"error": f"strong-scaling phase failed or timed out (rc {rc2}); the headline value is unaffected"}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(seconds=12.0):
    """Reference-faithful CPU loop (variant (i) of SURVEY 8d): 1 env, batch-1 MLP forward per
    step, 256-step buffer, GAE + 5 full-batch Adam steps -- the oracle ('port'), rank 0 only."""
    import numpy as np
    import torch
    from oracle import ppo_oracle as po
    from oracle.env_oracle import OracleEnv
    # batch-1 forwards and 256-sample updates do not scale with threads; the box's default (128 threads on a
    # 16-CPU share) oversubscribes and runs this loop ~6x SLOWER than one thread -- time the fast setting
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(0)
    p = {}
    shapes = {"feature.0.weight": (256, 6), "feature.0.bias": (256,), "feature.1.weight": (256,), "feature.1.bias": (256,),
              "feature.3.weight": (128, 256), "feature.3.bias": (128,), "feature.4.weight": (128,), "feature.4.bias": (128,),
              "actor.weight": (5, 128), "actor.bias": (5,), "critic.weight": (1, 128), "critic.bias": (1,)}
    for k, s in shapes.items():
        if k.endswith("weight") and len(s) == 2:
            w = torch.empty(s)
            torch.nn.init.orthogonal_(w, gain=0.01 if k.startswith("actor") else (1.0 if k.startswith("critic") else 2 ** 0.5),
                                      generator=gen)
            p[k] = w
        else:
            p[k] = torch.ones(s) if k in ("feature.1.weight", "feature.4.weight") else torch.zeros(s)
    adam = po.AdamState(p)
    env = OracleEnv("v2.0", seed=0)
    state = env.obs()
    buf = {k: [] for k in "sarvld"}
    steps = updates = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        with torch.no_grad():
            probs, value, _ = po.mlp_forward(p, torch.from_numpy(state)[None])
            a = int(torch.multinomial(probs[0], 1, generator=gen))
            lp = float(po.categorical_logp(probs, torch.tensor([a])))
        o, r, d, s, _ = env.step(a)
        for k, x in zip("sarvld", (state, a, r, float(value), lp, float(d))):
            buf[k].append(x)
        steps += 1
        if len(buf["s"]) >= 256:
            po.update_model(p, adam, np.stack(buf["s"]), np.array(buf["a"]), np.array(buf["r"], np.float32),
                            np.array(buf["v"], np.float32), np.array(buf["l"], np.float32), np.array(buf["d"], np.float32))
            buf = {k: [] for k in buf}
            updates += 1
        state = env.reset() if d else o
    dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} env steps of 1 env + {updates} _update_model calls (256-sample buffer, MLP policy, "
                      f"{dt:.1f} s) -- reference-faithful CPU loop of oracle/",
            "updates_per_s": updates / dt}


def host_cores(cap=16):
    """Host threads this process may actually use: the affinity mask, cut by the cgroup CPU quota when there is one,
    and by `cap` (a one-GPU box's CPU share is 16; more torch threads than that oversubscribe and run slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline_vectorised(T=128, H=128, n_sample=4096, n_full=4096, epochs=5, bank_fields=8):
    """Variant (ii) of SURVEY 8d: the SAME iteration as the GPU run (vectorised envs, LSTM(h) actor-critic, GAE,
    whole-buffer normalisation, `epochs` full-batch Adam steps) in numpy / torch-CPU batch operations on all host
    cores.  Bounded sample: n_sample of the n_full envs for the full T steps and all epochs (CPU cost per env-step
    does not depend on N at these sizes); fields from a materialised bank built outside the timed region."""
    import numpy as np
    import torch
    from oracle import ppo_oracle as po
    from oracle.env_oracle import FieldBank
    from oracle.vec_env_oracle import NumpyVecEnv
    threads = host_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    N = n_sample
    bank = FieldBank.from_seed(bank_fields, "v2.0", seed=0)
    env = NumpyVecEnv(N, bank, "v2.0")
    p = po.init_lstm_policy(6, H, 1, seed=0)
    adam = po.AdamState(p)
    gen = torch.Generator().manual_seed(0)
    rng = np.random.RandomState(0)
    obs = env.reset()
    h = torch.zeros(1, N, H)
    c = torch.zeros(1, N, H)
    t0 = time.perf_counter()
    # ---- rollout (train_ppo2.0.py:157-198 for N envs)
    B = {k: [] for k in ("obs", "act", "rew", "val", "logp", "done", "keep")}
    keep = np.ones(N, np.float32)
    h0, c0 = h.clone(), c.clone()
    with torch.no_grad():
        for t in range(T):
            probs, value, _, (h, c) = po.lstm_policy_forward(p, torch.from_numpy(obs)[None], h, c,
                                                             keep=torch.from_numpy(keep)[None])
            a = torch.multinomial(probs[0], 1, generator=gen).squeeze(1)
            lp = po.categorical_logp(probs[0], a)
            B["obs"].append(obs)
            B["act"].append(a.numpy())
            B["val"].append(value[0].numpy())
            B["logp"].append(lp.numpy())
            B["keep"].append(keep)
            obs, rew, done, _, _, _ = env.step(a.numpy(), rng.randn(N, 2))
            B["rew"].append(rew.astype(np.float32))
            B["done"].append(done.astype(np.float32))
            keep = 1.0 - done.astype(np.float32)
    t_roll = time.perf_counter() - t0
    S = {k: np.stack(v) for k, v in B.items()}          # time-major [T, N, ...]
    # ---- GAE (train_ppo2.0.py:18-32), vectorised over the envs
    rew, val, dn = S["rew"], S["val"], S["done"]
    adv = np.zeros((T, N), np.float32)
    last = np.zeros(N, np.float32)
    g, gl = np.float32(0.99), np.float32(0.99 * 0.95)
    for t in range(T - 1, -1, -1):
        nnt = 1 - (dn[t] if t == T - 1 else dn[t + 1])
        nv = (val[t] if t == T - 1 else val[t + 1]) * nnt
        last = (rew[t] + g * nv - val[t]) + gl * nnt * last
        adv[t] = last
    adv_n, ret = po.normalise(adv, val)
    x = torch.from_numpy(S["obs"])
    k = torch.from_numpy(S["keep"])
    act, lp_old, v_old = (torch.from_numpy(S[n]).reshape(-1) for n in ("act", "logp", "val"))
    # ---- EPOCHS full-batch optimiser steps (train_ppo2.0.py:43-88)
    for _ in range(epochs):
        leaf = {n: v.detach().clone().requires_grad_(True) for n, v in p.items()}
        probs, value, _, _ = po.lstm_policy_forward(leaf, x, h0, c0, keep=k)
        total, _, _, _ = po.ppo_losses(probs.reshape(T * N, -1), value.reshape(-1), act, lp_old, adv_n, ret, v_old)
        total.backward()
        grads = {n: leaf[n].grad for n in p}
        po.clip_grads(grads)
        adam.step(p, grads)
    dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    return {"value": N * T / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"ONE iteration of {N} of the {n_full} envs x {T} steps (LSTM h={H} rollout {t_roll:.1f} s + GAE + "
                      f"{epochs} full-batch epochs, {dt:.1f} s) in numpy / torch-CPU batch ops, {threads} threads; "
                      f"materialised bank of {bank_fields} fields built outside the timed region",
            "rollout_env_steps_per_s": N * T / t_roll, "seconds": dt}


# ------------------------------------------------------------------------------------------------ measurement
def measure(tr, steps, warmup, world, dev, ops, dist, timers=(), dominant=("lstm_bwd",), breakdown_iters=3):
    """`warmup` untimed iterations, then exactly `steps` timed ones between barrier + synchronize pairs;
    returns (max-over-ranks seconds, rollout ms per iteration, kernel timers).

    Inside the timed region only the DOMINANT kernel's launches are bracketed with HIP events on the launch stream (what
    `roofline` needs, live): an event pair costs ~5 us of device idle per bracketed launch (rocprofv3 trace of round 5:
    profiles/r05_trace_gaps_*.txt -- 20 brackets were 0.11 ms of a 7 ms C3 iteration, 7 % of a C2 iteration).  The other kernels
    of `timers` and the rollout are timed the same way over `breakdown_iters` EXTRA iterations after the timed region."""
    import torch

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        tr.train_iteration()
    live = tuple(t for t in timers if t in dominant)
    if live:
        ops.KERNEL_TIMER.enable(live)
    barrier()
    t0 = time.perf_counter()
    verbose = os.environ.get("UAV_BENCH_VERBOSE")
    if verbose == "trace":
        import faulthandler
        faulthandler.dump_traceback_later(2.0, repeat=True, file=sys.stderr)
    for k in range(steps):
        tk = time.perf_counter()
        tr.train_iteration()        # the function users call: rollout, update, curriculum, the range guard's poll
        if verbose:
            print(f"[bench] step {k}: host {1e3 * (time.perf_counter() - tk):.1f} ms", file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    if verbose == "trace":
        faulthandler.cancel_dump_traceback_later()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    summary = ops.KERNEL_TIMER.summary() if live else {}
    ops.KERNEL_TIMER.disable()
    # per-kernel breakdown + rollout time: a few more iterations of the same loop, outside the timed region
    rest = tuple(t for t in timers if t not in live)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(breakdown_iters)]
    if rest:
        ops.KERNEL_TIMER.enable(rest)
    tr.time_rollouts(ev)           # an event pair around each of those iterations' rollouts, recorded inside train_iteration()
    for _ in range(breakdown_iters):
        tr.train_iteration()
    barrier()
    if rest:
        summary.update(ops.KERNEL_TIMER.summary())
        ops.KERNEL_TIMER.disable()
    tr.losses()     # raises (on every rank) if any NaN probability was seen (reference convention)
    roll_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    return dt, roll_ms, summary


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, sorted by name): what a PMC profile was taken ON.  (git is not
    available on the GPU box -- the snapshot has no .git -- so staleness is decided by content, not by commit.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(tag_glob="r0*_hbm_traffic_pmc.json"):
    """Newest committed PMC summary under profiles/ (a profiler cannot run inside the timed region).  `stale` is True when
    the kernel sources have changed since the profile was taken (its `_meta.csrc_sha` != today's digest; profiles from
    before round 3 carry no digest and count as stale): such counter bytes are NOT used for the headline fraction."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", tag_glob)))
    if not files:
        return None, None
    f = files[-1]
    d = json.load(open(f))
    meta = d.get("_meta", {})
    now = csrc_digest()
    return d, {"file": os.path.relpath(f, ROOT), "git_commit": meta.get("git_commit"), "csrc_sha": meta.get("csrc_sha"),
               "csrc_sha_now": now, "stale": meta.get("csrc_sha") != now, "shape": meta.get("shape"),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE x2 (gfx950 "
                         "correction, MI355X_MICROARCH.md HBM), KB -> bytes, averaged per launch"}


def load_sq_counters(which="c3"):
    """Newest committed SQ-counter summary (tools/profile_pmc_sq.sh -> profiles/r0*_pmc_sq_<which>.json): per kernel
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs) and the SQ wait / active fractions.
    Same staleness rule as the byte counters (`_meta.csrc_sha` against today's kernel sources)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_sq_%s.json" % which)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    meta = d.get("_meta", {})
    now = csrc_digest()
    return d, {"file": os.path.relpath(files[-1], ROOT), "csrc_sha": meta.get("csrc_sha"), "csrc_sha_now": now,
               "stale": meta.get("csrc_sha") != now,
               "calibration": "profiles/r04_b_pmc_sq_calibration.json: a saturated fp16 MFMA loop reads 0.945 at 2.03 PFLOP/s"}


def binding_roof(bytes_moved, executed_flops, sec, flop_peak_tflops):
    """The roof that actually binds: the larger of (bytes moved / 8 TB/s) and (EXECUTED matrix flops / the peak of the
    pipe they run on).  executed_flops may be a list of (flops, peak TFLOP/s) pairs for work spread over two pipes: the
    matrix-pipe fraction is then the sum of the times each part needs at its own peak, and `peak` the blended rate.
    Returns the top-level roofline fields."""
    hbm = bytes_moved / sec / 1e9
    if isinstance(executed_flops, (list, tuple)):
        total = sum(f for f, _ in executed_flops)
        t_peak = sum(f / (pk * 1e12) for f, pk in executed_flops)
        executed_flops, flop_peak_tflops = total, total / t_peak / 1e12
    mf = executed_flops / sec / 1e12
    if hbm / PEAK_HBM_GBPS >= mf / flop_peak_tflops:
        return {"bound": "hbm", "achieved": hbm, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBPS}
    return {"bound": "mfma", "achieved": mf, "peak": flop_peak_tflops, "unit": "TFLOP/s", "frac": mf / flop_peak_tflops}


def roofline_block(cfg, N, T, H, timers, dt_iter, epochs, reused_fwd, kind, arith="fp16x3"):
    """Dominant kernel + whole iteration.  Top level (`bound`, `achieved`, `peak`, `unit`, `frac`) of the dominant kernel =
    SURVEY 8(d)'s algorithmic figure (f32-equivalent flops per launch / live duration / the dense f32 MFMA peak), `traffic` =
    HBM bytes per launch by the rocprofv3 counters (when the committed profile matches today's kernel sources, else null).
    What the implementation really moves and executes rides along: `hbm.frac_pmc` (counter bytes against 8 TB/s),
    `mfma.frac_executed_pipe` (three fp16 products per f32 product on the 2.5 PF pipe), `binding` (the larger of the two:
    the roof that actually binds the launch), and the same for the whole iteration under `iteration`."""
    L = max(cfg["layers"], 1)
    units = N * T
    traffic, src = load_traffic()
    fresh = bool(traffic) and not src["stale"] and cfg is CONFIGS["c3"]
    # executed products per f32 product and the pipe they run on
    mult, pipe_peak, pipe = {"fp16x3": (3, PEAK_F16_MFMA_TFLOPS, "fp16 MFMA, 3 piece products per f32 product"),
                             "bf16x6": (6, PEAK_F16_MFMA_TFLOPS, "bf16 MFMA, 6 piece products per f32 product"),
                             "f32": (1, PEAK_F32_MFMA_TFLOPS, "exact-f32 MFMA")}[arith]
    out = None
    bwd = timers.get("lstm_bwd")
    if bwd and kind == "lstm" and L == 1 and H in (64, 128):
        sec = bwd["avg_ms"] * 1e-3
        kname = "lstm_bwd_h3k_kernel<%d>" % H
        alg_b = ALG_UPDATE_B * units                                   # 8(d): 44 B read per unit per epoch
        impl_b = units * (5 * H + 4 * H + NHEADS + 1) * 4              # stash 5H read + dgates 4H written + dheads + keep
        fl = units * 2 * 4 * H * H                                     # dh_{t-1} = dG W_hh: one forward-equivalent product
        pmc = traffic.get(kname, {}).get("hbm_total_bytes") if fresh else None
        moved = pmc if pmc else impl_b
        # Top level = SURVEY 8(d)'s figure for this kernel, as the bench contract words it: ALGORITHMIC flops per launch (2 * 4H * H
        # per env-step, f32-equivalent: one forward-equivalent product) / the live launch duration, against the roof 8(d) names
        # for the LSTM GEMMs, the dense f32 MFMA peak.  `traffic` = HBM bytes per launch by the counters.  What the launch
        # actually moves / executes -- counter bytes against 8 TB/s (`hbm.frac_pmc`), three fp16 piece products per f32 product
        # against the 2.5 PF pipe (`mfma.frac_executed_pipe`) -- rides along, as does `binding` (the larger of those two).
        out = {"kernel": kname, "bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": fl / sec / 1e12 / PEAK_F32_MFMA_TFLOPS}
        out.update({"traffic": pmc, "traffic_source": src, "bytes_basis": "pmc" if pmc else "formula (stash 5H read + dgates 4H written + dheads + keep)",
                    "bytes_per_launch": moved, "avg_ms": bwd["avg_ms"], "launches": bwd["n"],
                    "algorithmic_flops_per_launch": fl, "algorithmic_flops_per_unit": 2 * 4 * H * H, "units_per_launch": units,
                    "note": "frac = SURVEY 8(d) algorithmic f32-equivalent flops per launch / live HIP-event duration on the launch stream / the dense "
                            "f32 MFMA peak (MI355X_MICROARCH.md).  The kernel EXECUTES those products as three fp16 piece products on the 2.5 PF "
                            "pipe (mfma.frac_executed_pipe) and streams its BPTT stash (hbm.frac_pmc = counter bytes / duration / 8 TB/s; "
                            "hbm.pmc_over_algorithmic = counter bytes over 8(d)'s 44 B per unit)",
                    "binding": binding_roof(moved, mult * fl, sec, pipe_peak),
                    "mfma": {"algorithmic_f32_tflops": fl / sec / 1e12, "f32_peak_tflops": PEAK_F32_MFMA_TFLOPS,
                             "frac_f32_peak": fl / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                             "executed_tflops": mult * fl / sec / 1e12, "executed_pipe": pipe, "executed_pipe_peak_tflops": pipe_peak,
                             "frac_executed_pipe": mult * fl / sec / 1e12 / pipe_peak},
                    "hbm": {"algorithmic_bytes": alg_b, "implementation_bytes": impl_b, "implementation_over_algorithmic": impl_b / alg_b,
                            "achieved_GBps_algorithmic": alg_b / sec / 1e9, "achieved_GBps_implementation": impl_b / sec / 1e9,
                            "peak_GBps": PEAK_HBM_GBPS, "frac_algorithmic": alg_b / sec / 1e9 / PEAK_HBM_GBPS,
                            "frac_implementation": impl_b / sec / 1e9 / PEAK_HBM_GBPS,
                            "frac_pmc": (pmc / sec / 1e9 / PEAK_HBM_GBPS) if pmc else None,
                            "pmc_over_algorithmic": (pmc / alg_b) if pmc else None}})
        # north_star: "evidenced by rocprof ... MFMA utilisation": the matrix pipe's busy fraction from the SQ counters, next to
        # the analytic one (executed flops / duration / peak).  Measured under the profiler, so at the profiler's clock.
        sq, sq_src = load_sq_counters("c3" if cfg is CONFIGS["c3"] else "none")
        if sq and kname in sq:
            e = sq[kname]
            out["mfma"].update({"counter_mfma_busy_frac": e.get("mfma_busy_frac"), "counter_wait_any_frac": e.get("wait_any_frac"),
                                "counter_wait_inst_any_frac": e.get("wait_inst_any_frac"), "counter_active_inst_any_frac": e.get("active_inst_any_frac"),
                                "counter_source": sq_src,
                                "counter_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); counts every MFMA of the kernel (the "
                                                "fp16 piece products AND the two exact-f32 MFMAs per step that form dy), hence a little above the "
                                                "analytic fp16-only fraction"})
    # whole iteration (per GPU): 8(d) per env-step figures x units, against the measured iteration time
    I = 6 + cfg.get("trend_k", 0)
    if kind == "lstm":
        fwd_fl = 2 * 4 * H * (I + H) + (L - 1) * 2 * 4 * H * (H + H) + 2 * H * NHEADS
        # implementation bytes: per forward pass (the rollout's included) gates 4H + c_prev H + y H written -- with I <= 6 the
        # stash's h_prev slot is not written at all (the weight gradients take h_prev from y); wider inputs (C5) write it and
        # the fp16 piece planes of y as well: 8H -- then per epoch bwd 5H read + 4H written, wgrad 4H + H read, all f32
        fw = 6 * H if I <= 6 else 8 * H
        impl_it = units * 4 * L * (fw + (epochs - (1 if reused_fwd else 0)) * fw + epochs * (9 * H + 5 * H)) \
            + units * (ALG_ROLLOUT_B + ALG_GAE_B + epochs * ALG_UPDATE_B)
        it_mult, it_peak = mult, pipe_peak
    else:
        fwd_fl = 2 * (I * 256 + 256 * 128 + 128 * NHEADS)
        impl_it = units * (ALG_ROLLOUT_B + ALG_GAE_B + epochs * ALG_UPDATE_B)     # the fused MLP kernels keep everything else on chip
        it_mult, it_peak = 1, PEAK_F32_MFMA_TFLOPS
    alg_it = units * (ALG_ROLLOUT_B + ALG_GAE_B + epochs * ALG_UPDATE_B)
    fl_it = units * fwd_fl * (1 + 3 * epochs)
    pm = None
    if fresh:
        def tb(prefix):
            return sum(v["hbm_total_bytes"] for k, v in traffic.items() if k.startswith(prefix))
        pm = tb("rollout_lstm_kernel") + (epochs - (1 if reused_fwd else 0)) * tb("lstm_fwd_h3_kernel") \
            + epochs * (tb("lstm_bwd_h3k_kernel") + tb("lstm_wgrad_h3_kernel") + tb("ppo_loss_kernel") + tb("wgrad_reduce_kernel"))
    if kind == "lstm":
        exec_it = it_mult * fl_it
    elif arith == "fp16x3":      # rollout: one forward; per epoch: forward + twice that backward (dW1's input gradient is not formed)
        exec_it = [(units * (MLP_F32_FWD * (1 + 3 * epochs) - epochs * 2 * 6 * 256), PEAK_F32_MFMA_TFLOPS),
                   (units * 3 * MLP_H3_FWD * (1 + 3 * epochs), PEAK_F16_MFMA_TFLOPS)]
    else:
        exec_it = fl_it
    whole = binding_roof(pm if pm else impl_it, exec_it, dt_iter, it_peak)
    whole.update({"bytes_basis": "pmc (big kernels)" if pm else "formula", "bytes_per_iteration": pm if pm else impl_it,
                  "algorithmic_bytes": alg_it, "algorithmic_GBps": alg_it / dt_iter / 1e9,
                  "frac_hbm_algorithmic": alg_it / dt_iter / 1e9 / PEAK_HBM_GBPS,
                  "implementation_bytes_formula": impl_it, "implementation_over_algorithmic": impl_it / alg_it,
                  "implementation_GBps": impl_it / dt_iter / 1e9,
                  "algorithmic_f32_tflops": fl_it / dt_iter / 1e12, "frac_f32_mfma_peak": fl_it / dt_iter / 1e12 / PEAK_F32_MFMA_TFLOPS,
                  "executed_tflops": (sum(f for f, _ in exec_it) if isinstance(exec_it, list) else exec_it) / dt_iter / 1e12})
    if isinstance(exec_it, list):
        whole["executed_by_pipe"] = [{"tflops": f / dt_iter / 1e12, "peak_tflops": pk, "frac": f / dt_iter / 1e12 / pk} for f, pk in exec_it]
        whole["frac_executed_pipe"] = sum(f / (pk * 1e12) for f, pk in exec_it) / dt_iter
    else:
        whole["executed_pipe_peak_tflops"] = it_peak
        whole["frac_executed_pipe"] = exec_it / dt_iter / 1e12 / it_peak
    if pm:
        whole["pmc_bytes_big_kernels"] = pm
        whole["pmc_over_algorithmic"] = pm / alg_it
        whole["pmc_GBps"] = pm / dt_iter / 1e9
    if out is None:
        out = {"kernel": None}
        out.update({k: whole[k] for k in ("bound", "achieved", "peak", "unit", "frac")})
        out.update({"traffic": None, "traffic_source": None, "bytes_basis": whole["bytes_basis"],
                    "note": "whole-iteration figures (this configuration has no single dominant sequence kernel timed live): the larger "
                            "of implementation bytes / 8 TB/s and executed matrix flops / the executing pipe's peak"})
    out["iteration"] = whole
    return out


def measure_exchanges(dev, n_params, own_group):
    """Per-call time of the iteration's three exchanges through the process group that is up (RCCL; with `own_group` a ONE-rank
    communicator brought up just for this, after the timed region): HIP events around 50 calls each, on the stream the
    collectives are issued on.  One rank measures what a collective costs to ISSUE and run with nobody to talk to -- the floor
    under the 8-rank latency, which only the driver's multi-GPU node can measure."""
    import torch
    import torch.distributed as dist
    out = {}
    try:
        if own_group:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        world = dist.get_world_size()
        grad = torch.zeros(n_params, dtype=torch.float32, device=dev)
        stats = torch.zeros(3, dtype=torch.float64, device=dev)
        msg = torch.zeros(4 + 16384 + 1, dtype=torch.uint8, device=dev)
        parts = [torch.empty_like(msg) for _ in range(world)]
        calls = {f"grad_allreduce_{n_params * 4 // 1024}KB": lambda: dist.all_reduce(grad),
                 "adv_stats_allreduce_3_doubles": lambda: dist.all_reduce(stats),
                 "success_allgather_16KB": lambda: dist.all_gather(parts, msg)}
        for name, fn in calls.items():
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) * 1e3 / 50
        out["world"] = world
        out["backend"] = dist.get_backend()
        if own_group:
            dist.destroy_process_group()
    except Exception as e:   # the probe must never cost the bench its line
        out["error"] = repr(e)[:200]
    return out


def measure_exchanges_isolated(local, n_params, timeout=90.0):
    """The one-rank RCCL probe in a FRESH child process with a time limit: bringing up a communicator is the one step of this
    program that talks to a runtime outside the build, and the bench line must not depend on it.  The child touches the GPU
    itself (nothing is exec'ed from this process's HIP state); on expiry it is killed by PID and the probe reports the fact."""
    code = ("import json, os, sys\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "import torch, bench\n"
            f"torch.cuda.set_device({int(local)})\n"
            f"r = bench.measure_exchanges(torch.device('cuda', {int(local)}), {int(n_params)}, own_group=True)\n"
            "print('PROBE ' + json.dumps(r), flush=True)\n")
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
        for line in p.stdout.splitlines():
            if line.startswith("PROBE "):
                return json.loads(line[6:])
        return {"error": f"probe child exited {p.returncode} without a result: {p.stderr[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"probe child killed after {timeout:.0f} s"}
    except Exception as e:
        return {"error": repr(e)[:200]}


def predicted_8gpu(cfg_name, cfg, N, T, ms_iter, n_params, measured=None):
    """What the first 8-GPU SCALE record should show, from THIS run's one-GPU iteration time: weak scaling =
    8 x the per-GPU work in (t_1 + the iteration's exchanges); the exchanges are 5 all-reduces of the flat gradient
    (latency-bound over xGMI at 8 ranks: a RANGE of 50-150 us each, 120-300 us at C5's 3.2 MB; never measured, no 8-GPU node)
    + one 3-double all-reduce + one 16 KB all-gather on a side stream (30-90 us together, the all-gather overlapped).
    `measured_one_rank_us`: what each exchange costs through a one-rank RCCL communicator on this GPU (the issue floor)."""
    small = n_params * 4 < (1 << 20)
    lo, hi = (50.0, 150.0) if small else (120.0, 300.0)
    t_lo, t_hi = (5 * lo + 30.0) * 1e-3, (5 * hi + 90.0) * 1e-3
    one = N * T / (ms_iter * 1e-3)
    w_hi, w_lo = 8 * N * T / ((ms_iter + t_lo) * 1e-3), 8 * N * T / ((ms_iter + t_hi) * 1e-3)
    out = {"from_ms_per_step_1gpu": ms_iter, "assumed_exchange_ms_per_iteration": [t_lo, t_hi],
           "assumed_us_per_gradient_allreduce_8_ranks": [lo, hi],
           "weak_8gpu_env_steps_per_s": [w_lo, w_hi], "weak_8gpu_over_1gpu": [w_lo / one, w_hi / one],
           "note": "sequence kernels take T x (one workgroup's step latency) whatever the number of workgroups up to one per CU, "
                   "so splitting a config's envs over more GPUs (strong scaling) leaves the iteration time nearly flat"}
    if measured:
        out["measured_one_rank_us"] = measured
    if cfg_name == "c4":
        out["vs_one_gpu_holding_all_8192_envs"] = ("one GPU with 8192 envs runs two 16-env tiles per CU, ~2 x the C3 iteration (~14 ms, ~75 M "
                                                   "env-steps/s): 8 GPUs at 1024 envs each are predicted ~2.8-3 x that, NOT north_star's >= 6 x")
    return out


def distributed_block(world, rank, local, dev, args, dist):
    """What the communicator saw, for the first multi-GPU record: world size, backend, RCCL's version as torch reports it (and as
    the C ABI's dlopen'ed copy does), and every rank's (rank, local rank, device index, device name, PCI bus id) gathered to
    rank 0 -- N distinct devices = N ranks each on its own GPU.  Every rank calls this (the gather is a collective)."""
    import torch
    me = {"rank": rank, "local_rank": local, "device_index": dev.index, "pid": os.getpid()}
    try:
        pr = torch.cuda.get_device_properties(dev)
        me["device_name"] = pr.name
        me["pci_bus_id"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
        me["uuid"] = str(getattr(pr, "uuid", ""))
    except Exception as e:
        me["device_error"] = repr(e)[:120]
    ranks = [me]
    if dist.is_initialized() and world > 1:
        got = [None] * world
        dist.all_gather_object(got, me)
        ranks = got
    out = {"world_size": world, "initialized": bool(dist.is_initialized()),
           "backend": dist.get_backend() if dist.is_initialized() else None, "collectives": args.collectives,
           "pg_timeout_s": args.pg_timeout, "ranks": ranks,
           "distinct_devices": len({(r.get("pci_bus_id"), r.get("uuid"), r.get("device_index")) for r in ranks})}
    try:
        out["rccl_version_torch"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:
        out["rccl_version_torch"] = repr(e)[:120]
    try:
        from uavppo import ops
        out["rccl_version_abi"] = ops.rccl_version()
    except Exception as e:
        out["rccl_version_abi"] = repr(e)[:120]
    return out


def build_trainer(cfg, N, rank, world, dev, ops):
    import torch
    from uavppo.trainer import VecPPOTrainer
    bank = bank_src = None
    if cfg.get("bank_fields"):
        # synthetic bank generated with E3's formula by the procedural sampler itself (env_materialise kernel)
        from uavppo.vec_env import VecMethaneEnv
        F = cfg["bank_fields"]
        gen = VecMethaneEnv(F, cfg["variant"], dev, seed=4321)
        gen.reset()
        bank = torch.stack([ops.env_materialise(gen.state, F, gen.cfg(), f) for f in range(F)])
        bank_src = gen.peek()[1]
    kind = cfg.get("policy", "lstm")
    return VecPPOTrainer(N, cfg["horizon"], kind, hidden=cfg["hidden"] or 128, layers=cfg["layers"] or 1,
                         variant=cfg["variant"], device=dev, seed=1234, rank=rank, world_size=world, bank=bank,
                         bank_sources=bank_src, trend_k=cfg.get("trend_k", 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-phase", action="store_true", help="launcher mode: skip the second (strong-scaling) set of ranks")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-exchange-probe", action="store_true", help="skip timing the iteration's collectives after the timed region")
    ap.add_argument("--collectives", default="torch", choices=("torch", "abi"),
                    help="carrier of the iteration's exchanges: torch.distributed (default) or the C ABI's own RCCL communicator (uav_allreduce)")
    ap.add_argument("--pg-timeout", type=float, default=600.0, help="seconds before a stuck collective (or a rendezvous nobody joins: a fresh box can take 1-2 min to load PyTorch) aborts the rank")
    ap.add_argument("--headline-timeout", type=float, default=0.0, help="launcher mode: seconds before the headline rank set is terminated (0 = config-dependent default)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above touched the GPU (no torch import yet),
        # every rank is a fresh process; rank 0 prints the one JSON line.
        raise SystemExit(launch(args, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON.  Native libraries write there too (RCCL prints a version banner on stdout
    # when its first communicator comes up), so from here on fd 1 is stderr and the line goes out through the saved fd.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one GPU (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # UAVPPO_FORCE_COLLECTIVES=1 at one rank: every exchange of the iteration goes through a one-rank communicator
    # (RCCL rehearsal on a one-GPU box: what the collectives' launches cost, and that the code path runs at all)
    rehearsal = world == 1 and os.environ.get("UAVPPO_FORCE_COLLECTIVES") == "1"
    if world > 1 or rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        # a collective that never completes (a dead or missing peer) must end the run with a non-zero exit code instead of
        # sitting in the launcher: the process group's watchdog aborts the process after `--pg-timeout` seconds
        import datetime
        pg_timeout = datetime.timedelta(seconds=args.pg_timeout)
        if args.backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=pg_timeout)
        if args.collectives == "abi":
            # the three exchanges on the C ABI's own RCCL communicator (uav_allreduce & co, include/uavppo.h K9); the torch
            # group above then only carries the 128-byte id, the barriers and the max-over-ranks time
            from uavppo import dist_utils
            dist_utils.use_abi_collectives(rank, world, dev)

    from uavppo import ops
    cfg = CONFIGS[args.config]
    kind = cfg.get("policy", "lstm")
    T, H = cfg["horizon"], cfg["hidden"]
    n_cfg = cfg["num_envs"]
    if n_cfg % world and args.scaling == "strong":
        raise SystemExit(f"config {args.config}: {n_cfg} envs do not split over {world} ranks")
    n_weak, n_strong = n_cfg, n_cfg // world
    N = n_weak if args.scaling == "weak" else n_strong
    tr = build_trainer(cfg, N, rank, world, dev, ops)
    dt, roll_ms, timers = measure(tr, args.steps, args.warmup, world, dev, ops, dist,
                                  timers=("lstm_bwd", "lstm_fwd", "lstm_wgrad", "rollout", "ppo_loss"))
    epochs = tr.hp["epochs"]
    env_steps = N * T * world * args.steps
    value = env_steps / dt
    opt_steps = epochs * tr.num_minibatches
    reused = kind == "lstm" and cfg["layers"] == 1 and H in (64, 128) and not cfg.get("trend_k")
    arith = {"fp16x3": "fp16x3", "bf16x6": "bf16x6"}.get(getattr(tr, "arith", "fp16x3"), "fp16x3")
    if os.environ.get("UAV_LSTM_F32_MFMA"):
        arith = "f32"
    roofline = roofline_block(cfg, N, T, H, timers, dt / args.steps, epochs, reused, kind, arith)
    pol = f"LSTM h={H} x{cfg['layers']}" if kind == "lstm" else "MLP 6-256-128 (the reference's policy)"
    out = {
        "metric": f"env-steps/sec (rollout + GAE + {epochs}-epoch PPO update), {N} envs x {T} T per GPU, {pol}",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (procedural Gaussian-plume envs, random-init actor-critic)",
        "arithmetic": "f32 results throughout; the recurrent LSTM products are evaluated as three fp16 piece products per f32 "
                      "product (two-piece operand split carrying 24 bits, gradients block-scaled by powers of two), weight gradients "
                      "included, all with f32 accumulation -- error vs f64 no larger than the "
                      "exact-f32 MFMA chain's (tests/test_gpu_lstm.py::test_split_kernels_have_f32_accuracy); env arithmetic in f64",
        "config": {"workload": f"BASELINE config {args.config.upper()}: PPO{cfg['variant'].upper()}, {N} envs/GPU x T={T}, "
                               f"{pol}, {'materialised bank F=%d' % cfg['bank_fields'] if cfg.get('bank_fields') else 'procedural field'}, obs {6 + cfg.get('trend_k', 0)}, "
                               f"5 actions, reference_exact GAE, {epochs} epochs x {tr.num_minibatches} minibatch "
                               f"of {N * T} samples/GPU", "num_envs_per_gpu": N, "num_envs_total": N * world, "horizon": T, "hidden": H,
                   "minibatch_samples": N * T // tr.num_minibatches, "parallelism": f"dp{world} (env shards, RCCL grad all-reduce)"},
        "rollout_env_steps_per_s": N * T * world / (roll_ms * 1e-3),
        "ppo_iterations_per_s": args.steps / dt, "optimizer_steps_per_s": args.steps * opt_steps / dt,
        "rollout_ms": roll_ms, "kernel_ms": {k: v["avg_ms"] for k, v in timers.items()},
        "kernel_ms_note": "lstm_bwd (the roofline's kernel): HIP events around its launches INSIDE the timed region; the other kernels "
                          "and rollout_ms: the same brackets over 3 extra iterations after it (an event pair costs ~5 us of device idle)",
        "roofline": roofline,
    }
    # (the strong-scaling shape is NOT timed in these processes: launch() runs it in a second set of fresh ranks; under
    #  torchrun ask for it with --scaling strong)
    if world == 1:
        measured = None
        if not args.no_exchange_probe:
            measured = (measure_exchanges(dev, tr.policy.num_params(), own_group=False) if dist.is_initialized()
                        else measure_exchanges_isolated(local, tr.policy.num_params()))
        out["predicted_8gpu"] = predicted_8gpu(args.config, cfg, N, T, dt / args.steps * 1e3, tr.policy.num_params(), measured)
    elif not args.no_exchange_probe:
        out["measured_exchange_us"] = measure_exchanges(dev, tr.policy.num_params(), own_group=False)
    if rehearsal:
        out["rehearsal"] = f"one-rank {args.backend} communicator, all exchanges issued (UAVPPO_FORCE_COLLECTIVES=1)"
    out["distributed"] = distributed_block(world, rank, local, dev, args, dist)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline_vectorised"] = cpu_baseline_vectorised(T=128, H=128, n_full=4096)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
