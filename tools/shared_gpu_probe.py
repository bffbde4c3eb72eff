#!/usr/bin/env python3
"""What do SEVERAL processes sharing ONE GPU cost a kernel?  (The round-2 rehearsal `bench.py --gpus 2 --backend gloo` on a
one-GPU box showed a 100x slow second phase: gpurun_out/b2_c5.*, b2e.json.)  K fresh processes, no collectives at all, each
times single launches of the fused C3-strong rollout (2048 envs x 128 steps, ONE kernel, ~0.7 ms alone) with HIP events:

  phase A   every process launches back to back, all at once           -> how does the GPU multiplex process contexts?
  phase B   the same after each process built and dropped a 4096-env trainer first (the weak -> strong sequence)
  phase C   phase A with a host sync after every launch (the pattern a gloo all-reduce of GPU tensors imposes)

    python tools/shared_gpu_probe.py [K=2] [launches=150]
Prints per process and phase: min / median / p99 / max event time in ms, and the wall time of the phase."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(idx, K, n):
    sys.path.insert(0, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd"))
    import torch
    from uavppo.trainer import VecPPOTrainer
    dev = torch.device("cuda", 0)

    def phase(tag, tr, sync_each):
        for _ in range(5):
            tr.collect()
        torch.cuda.synchronize()
        # crude cross-process start line: the next multiple of 2 s on the wall clock
        time.sleep(2.0 - (time.time() % 2.0))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            tr.collect()
            b.record()
            if sync_each:
                b.synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"[probe] K={K} proc {idx} phase {tag}: min {ms[0]:.3f} med {ms[n // 2]:.3f} p99 {ms[int(n * 0.99)]:.3f} "
              f"max {ms[-1]:.3f} ms; wall {1e3 * wall / n:.3f} ms/launch", flush=True)

    tr = VecPPOTrainer(2048, 128, "lstm", hidden=128, device=dev, seed=1 + idx, use_curriculum=False)
    phase("A (back to back)", tr, False)
    phase("C (sync after each)", tr, True)
    del tr
    big = VecPPOTrainer(4096, 128, "lstm", hidden=128, device=dev, seed=9, use_curriculum=True)
    for _ in range(3):
        big.train_iteration()
    torch.cuda.synchronize()
    del big
    torch.cuda.empty_cache()
    tr = VecPPOTrainer(2048, 128, "lstm", hidden=128, device=dev, seed=1 + idx, use_curriculum=False)
    phase("B (after a dropped 4096-env trainer)", tr, False)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
        for k in sorted({1, K}):          # one process alone first: the reference timing
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(i), str(k), str(n)]) for i in range(k)]
            rc = max(p.wait() for p in ps)
            if rc:
                raise SystemExit(rc)
