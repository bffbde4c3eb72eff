"""bench.py's own rank launcher (plain `python bench.py --gpus N`): N fresh child processes with the
torch.distributed environment, one rendezvous, exit code propagation.  CPU: the children run a gloo all-reduce."""
import json
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
    sys.exit(3)                      # a rank dies before the next collective: the launcher must not hang
if len(sys.argv) > 1 and sys.argv[1] == "fail":
    dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": t.item()}), flush=True)
dist.destroy_process_group()
'''


def _launch(n, *args):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.spawn_ranks({n}, [sys.executable, '-c', {CHILD!r}] + {list(args)!r}, timeout=120))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_spawn_ranks_runs_one_job_and_rank0_prints_one_line():
    r = _launch(2)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "sum": 3.0}


def test_spawn_ranks_propagates_a_failing_rank_without_hanging():
    r = _launch(2, "fail")
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])


def test_bench_parent_spawns_before_importing_torch():
    """The launcher branch of bench.main() runs before anything that could initialise HIP: bench.py's module level
    imports neither torch nor the package."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    top_level = [l for l in head.splitlines() if l.startswith("import ") or l.startswith("from ")]
    assert not any("torch" in l or "uavppo" in l for l in top_level), top_level
    body = src[src.index("def main():"):]
    assert body.index("spawn_ranks(") < body.index("import torch")
