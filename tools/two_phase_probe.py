#!/usr/bin/env python3
"""Reproduce round 2's weak -> strong stall (gpurun_out/b2e.json, b2_c5.*) and split it: two gloo ranks sharing the one GPU
run the OLD in-process sequence -- trainer A (4096 envs), `del`, `empty_cache`, trainer B (2048 envs) -- at C3 (cheap:
milliseconds per iteration), timing the host side of collect / update / curriculum per iteration and the GPU side with
events.  Variants (argv[1], comma separated) remove one suspect at a time:

    base        the round-2 sequence
    notimer     phase A without ops.KERNEL_TIMER
    nocurr      both trainers without the curriculum (no side stream, no all-gather, no pinned ring)
    gc          gc.collect() + synchronize + barrier between the phases
    fresh       phase B only (what a fresh rank set runs)
    threads1    torch.set_num_threads(1) in every rank
    keepA       trainer A stays alive (no del, no empty_cache)
    noempty     del, but no torch.cuda.empty_cache()
    samestream  trainer B takes over trainer A's side stream instead of drawing a new one from torch's pool
    mainstream  trainer B packs / copies the success bits on the main stream (side_stream_curriculum = False)
    sameshape   phase A with 2048 envs as well

    python tools/two_phase_probe.py base,gc,fresh        # launcher: spawns 2 ranks per variant
"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]


def rank_main(variant):
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if variant == "threads1":
        torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uavppo import ops
    from uavppo.trainer import VecPPOTrainer

    keep = {}

    def phase(tag, n_env, steps, warm, timers):
        tr = VecPPOTrainer(n_env, 128, "lstm", hidden=128, device=dev, seed=1234, rank=rank, world_size=world,
                           use_curriculum=(variant != "nocurr"))
        if tag[0] == "B" and variant == "samestream":
            tr._side = keep["side"]
        if tag[0] == "B" and variant == "mainstream":
            tr.side_stream_curriculum = False
        keep["side"] = tr._side
        for _ in range(warm):
            tr.train_iteration()
        if timers:
            ops.KERNEL_TIMER.enable(("lstm_bwd", "lstm_fwd", "lstm_wgrad", "rollout", "ppo_loss"))
        dist.barrier()
        torch.cuda.synchronize()
        rows = []
        for k in range(steps):
            t0 = time.perf_counter()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            tr.collect()
            b.record()
            t1 = time.perf_counter()
            tr.update()
            t2 = time.perf_counter()
            tr.update_curriculum()
            t3 = time.perf_counter()
            tr.iteration += 1
            rows.append((t1 - t0, t2 - t1, t3 - t2, a, b))
        dist.barrier()
        torch.cuda.synchronize()
        if timers:
            ops.KERNEL_TIMER.summary()
            ops.KERNEL_TIMER.disable()
        tot = [1e3 * sum(r[i] for r in rows) / steps for i in range(3)]
        ev = sum(r[3].elapsed_time(r[4]) for r in rows) / steps
        worst = max(1e3 * (r[0] + r[1] + r[2]) for r in rows)
        print(f"[two-phase {variant}] rank {rank} {tag}: host ms/iter collect {tot[0]:.2f} update {tot[1]:.2f} curriculum {tot[2]:.2f} "
              f"| sum {sum(tot):.2f} worst {worst:.1f} | rollout event {ev:.2f} ms | threads {torch.get_num_threads()} "
              f"| mem {torch.cuda.memory_allocated() >> 20} MiB", flush=True)
        return tr

    if variant != "fresh":
        tr = phase("A 4096 envs", 2048 if variant == "sameshape" else 4096, 10, 3, variant != "notimer")
        if variant == "keepA":
            keep["A"] = tr
        del tr
        if variant not in ("keepA", "noempty"):
            torch.cuda.empty_cache()
        if variant == "gc":
            gc.collect()
            torch.cuda.synchronize()
            dist.barrier()
    phase("B 2048 envs", 2048, 10, 2, False)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ:
        rank_main(sys.argv[1])
    else:
        import bench
        for v in (sys.argv[1] if len(sys.argv) > 1 else "base,gc,fresh").split(","):
            rc, _ = bench.spawn_ranks(2, [sys.executable, os.path.abspath(__file__), v], timeout=240)
            print(f"[two-phase {v}] rc {rc}", flush=True)
            if rc:
                break
