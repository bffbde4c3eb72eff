"""Where does the one ~30 ms host stall around iteration 40 of a headline-configuration run come from?  Per-phase host times of
every iteration (collect / update / curriculum), garbage-collector passes with their durations, curriculum state."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo.trainer import VecPPOTrainer
tr = VecPPOTrainer(4096, 128, "lstm", hidden=128, device="cuda:0", seed=0)
gcs = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: gcs.append((tr.iteration, info["generation"], 1e3 * (time.perf_counter() - cb.t), info["collected"]))
gc.callbacks.append(cb)
for _ in range(3):
    tr.train_iteration()
torch.cuda.synchronize()
for it in range(60):
    t0 = time.perf_counter(); tr.collect()
    t1 = time.perf_counter(); tr.update()
    t2 = time.perf_counter(); r0 = tr.radius; tr.update_curriculum()
    t3 = time.perf_counter(); tr.iteration += 1
    if t3 - t0 > 0.012 or tr.radius != r0:
        print(f"iteration {tr.iteration}: collect {1e3 * (t1 - t0):.1f} ms  update {1e3 * (t2 - t1):.1f}  curriculum {1e3 * (t3 - t2):.1f}  radius {r0:.1f} -> {tr.radius:.1f} "
              f"bonus {type(tr.bonus).__name__}", flush=True)
torch.cuda.synchronize()
print("gc passes (iteration, generation, ms, collected):", [g for g in gcs if g[2] > 1.0 or g[1] == 2])
