#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec (+ PPO updates/sec) of the PPOV2.0 hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of BASELINE.json's headline config (C3): a fused
rollout of 4096 envs x 128 steps per GPU over the procedural Gaussian-plume environment with the
LSTM(h=128) actor-critic, the GAE scan + whole-buffer normalisation, 5 epochs of the clipped-PPO
update (forward, loss, BPTT, weight gradients, grad all-reduce, clip+Adam) and the curriculum
update.  Weak scaling: every rank owns 4096 envs; value = all ranks' env-steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {   # BASELINE.json configs (per-GPU env count; SURVEY 8 shorthand)
    "c2": dict(num_envs=256, horizon=64, hidden=64, layers=1, variant="v2.0"),
    "c3": dict(num_envs=4096, horizon=128, hidden=128, layers=1, variant="v2.0"),
    # C4: 8192 envs over 8 GPUs = 1024 per GPU, sigma=15 (PPOV2.1), materialised bank of F=64 fields in HBM
    "c4": dict(num_envs=1024, horizon=128, hidden=128, layers=1, variant="v2.1", bank_fields=64),
    # C5: 32768 envs over 8 GPUs = 4096 per GPU, T=256, LSTM h=256 stacked x2, obs 6 + 2 trend channels
    # (generic per-step LSTM path + step-wise rollout: correct, launch-bound)
    "c5": dict(num_envs=4096, horizon=256, hidden=256, layers=2, variant="v2.1", trend_k=2),
}
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA ~2.5 PFLOP/s (the split-fp16 kernels issue 3 products)
PEAK_HBM_GBS = 8000.0


def lstm_flops_per_env_step(I, H, A=5):
    """LSTM layer fwd 2*4H*(I+H); training = 3x fwd."""
    return 2 * 4 * H * (I + H)


def cpu_baseline(seconds=12.0):
    """Reference-faithful CPU loop (variant (i) of BASELINE.md 3): 1 env, batch-1 MLP forward per
    step, 256-step buffer, GAE + 5 full-batch Adam steps -- the oracle ('port'), rank 0 only."""
    import numpy as np
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = local % max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one GPU (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer
    cfg = CONFIGS[args.config]
    N, T, H = cfg["num_envs"], cfg["horizon"], cfg["hidden"]
    bank = bank_src = None
    if cfg.get("bank_fields"):
        # synthetic bank generated with E3's formula by the procedural sampler itself (env_materialise kernel)
        from uavppo.vec_env import VecMethaneEnv
        F = cfg["bank_fields"]
        gen = VecMethaneEnv(F, cfg["variant"], dev, seed=4321)
        gen.reset()
        bank = torch.stack([ops.env_materialise(gen.state, F, gen.cfg(), f) for f in range(F)])
        bank_src = gen.peek()[1]
    tr = VecPPOTrainer(N, T, "lstm", hidden=H, layers=cfg["layers"], variant=cfg["variant"], device=dev,
                       seed=1234, rank=rank, world_size=world, bank=bank, bank_sources=bank_src,
                       trend_k=cfg.get("trend_k", 0))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.train_iteration()
    # ---- timed region: exactly K iterations, barrier + synchronize on both sides
    ops.KERNEL_TIMER.enable(("lstm_bwd", "lstm_fwd", "lstm_wgrad", "rollout", "ppo_loss"))
    ev_roll = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev_roll[k][0].record()
        tr.collect()
        ev_roll[k][1].record()
        tr.update()
        tr.update_curriculum()
        tr.iteration += 1
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    timers = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER.disable()
    tr.losses()     # raises if any NaN probability was seen (reference convention)

    roll_ms = sum(a.elapsed_time(b) for a, b in ev_roll) / args.steps
    env_steps = N * T * world * args.steps
    value = env_steps / dt
    opt_steps = tr.hp["epochs"] * tr.num_minibatches
    # roofline of the dominant kernel, the BPTT sequence kernel lstm_bwd_h3k_kernel.  With its dh = dG W_hh product
    # on the fp16 matrix pipe (two-piece operand split, three products, f32 accuracy) it sits under the HBM roof, not the MFMA one:
    # algorithmic bytes per (env, step) = 5H stash values read + 4H gate gradients written + NH dheads + keep,
    # x N*T per launch (DESIGN.md "Kernels"), over its average launch duration timed live with HIP events on the
    # launch stream.  `traffic` = HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction +
    # WRITE_SIZE, separate passes), measured at this very shape and stored under profiles/ (a profiler cannot
    # run inside the timed region).  The MFMA side is reported next to it in f32-equivalent flops.
    NHEADS = 6
    bytes_bwd = N * T * (5 * H + 4 * H + NHEADS + 1) * 4 * cfg["layers"]
    fl_bwd = N * T * 2 * 4 * H * H * cfg["layers"]
    bwd = timers.get("lstm_bwd")
    roofline = None
    if bwd:
        sec = bwd["avg_ms"] * 1e-3
        ach = bytes_bwd / sec / 1e9
        traffic = None
        kname = "lstm_bwd_h3k_kernel<%d>" % H
        tf = os.path.join(ROOT, "profiles", "r01_j_hbm_traffic_pmc.json")
        if args.config == "c3" and os.path.exists(tf):
            traffic = json.load(open(tf)).get(kname, {}).get("hbm_total_bytes")
        roofline = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBPS, "traffic": traffic, "avg_ms": bwd["avg_ms"], "launches": bwd["n"],
                    "algorithmic_bytes": bytes_bwd,
                    "mfma": {"f32_equiv_tflops": fl_bwd / sec / 1e12, "executed_fp16_tflops": 3 * fl_bwd / sec / 1e12,
                             "fp16_peak_tflops": PEAK_BF16_MFMA_TFLOPS, "frac": 3 * fl_bwd / sec / 1e12 / PEAK_BF16_MFMA_TFLOPS}}
    out = {
        "metric": f"env-steps/sec (rollout + GAE + {tr.hp['epochs']}-epoch PPO update), {N} envs x {T} T per GPU, LSTM h={H}",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (procedural Gaussian-plume envs, random-init LSTM actor-critic)",
        "arithmetic": "f32 results throughout; the recurrent LSTM products are evaluated as three fp16 piece products per f32 "
                      "product (two-piece operand split carrying 24 bits, gradients block-scaled by powers of two), weight gradients "
                      "included, all with f32 accumulation -- error vs f64 no larger than the "
                      "exact-f32 MFMA chain's (tests/test_gpu_lstm.py::test_split_kernels_have_f32_accuracy); env arithmetic in f64",
        "config": {"workload": f"BASELINE config {args.config.upper()}: PPO{cfg['variant'].upper()}, {N} envs/GPU x T={T}, "
                               f"LSTM h={H} x{cfg['layers']}, {'materialised bank F=%d' % cfg['bank_fields'] if cfg.get('bank_fields') else 'procedural field'}, obs {6 + cfg.get('trend_k', 0)}, "
                               f"5 actions, reference_exact GAE, {tr.hp['epochs']} epochs x {tr.num_minibatches} minibatch "
                               f"of {N * T} samples/GPU", "num_envs_per_gpu": N, "horizon": T, "hidden": H,
                   "minibatch_samples": N * T // tr.num_minibatches, "parallelism": f"dp{world} (env shards, RCCL grad all-reduce)"},
        "rollout_env_steps_per_s": N * T * world / (roll_ms * 1e-3),
        "ppo_iterations_per_s": args.steps / dt, "optimizer_steps_per_s": args.steps * opt_steps / dt,
        "rollout_ms": roll_ms, "kernel_ms": {k: v["avg_ms"] for k, v in timers.items()},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
