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
            f"sys.exit(bench.spawn_ranks({n}, [sys.executable, '-c', {CHILD!r}] + {list(args)!r}, timeout=120)[0])")
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
    assert body.index("launch(args") < body.index("import torch")


def test_spawn_ranks_captures_rank0_stdout_and_times_out_a_hung_set():
    import bench
    rc, text = bench.spawn_ranks(2, [sys.executable, "-c", CHILD], timeout=120, capture_rank0=True)
    assert rc == 0 and bench.last_json_line(text) == {"n_gpus": 2, "sum": 3.0}
    hang = "import time, os\nif os.environ['RANK'] == '1':\n    time.sleep(600)\n"
    rc, text = bench.spawn_ranks(2, [sys.executable, "-c", hang], timeout=3, capture_rank0=True)
    assert rc == 124


FAKE_RANK = r'''
import json, os, sys, time
a = sys.argv[1:]
strong = a[a.index("--scaling") + 1] == "strong" if "--scaling" in a else False
if strong and os.environ.get("FAKE_STRONG") == "hang":
    time.sleep(600)
if not strong and os.environ.get("FAKE_STRONG") == "headline-hang":
    time.sleep(600)
if os.environ.get("FAKE_ENV_OUT") and os.environ["RANK"] == "0" and not strong:
    open(os.environ["FAKE_ENV_OUT"], "w").write(os.environ.get("GPU_MAX_HW_QUEUES", "unset"))
if strong and os.environ.get("FAKE_STRONG") == "fail":
    sys.exit(5)
if os.environ["RANK"] == "0":
    n = 128 if strong else 256
    print("banner noise")
    print(json.dumps({"value": 2.0 if strong else 1.0, "unit": "env-steps/s", "steps": 3, "ms_per_step": 1.5, "rollout_ms": 0.5,
                      "config": {"num_envs_per_gpu": n, "num_envs_total": 2 * n}}), flush=True)
'''


def _fake_launch(tmp_path, monkeypatch, mode, backend="gloo", cap=5):
    """bench.launch() with the rank command replaced by a stand-in: the launcher's own logic (two fresh rank sets, merge,
    a failing or hanging strong phase never loses the headline line) without a GPU."""
    import argparse
    import bench
    fake = tmp_path / "fake_rank.py"
    fake.write_text(FAKE_RANK)
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(fake))
    monkeypatch.setenv("FAKE_STRONG", mode)
    real = bench.spawn_ranks
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, cmd, **kw: real(n, cmd, **dict(kw, timeout=min(kw.get("timeout") or cap, cap))))
    args = argparse.Namespace(gpus=2, scaling="weak", no_strong_phase=False, steps=4, warmup=1, config="c2", backend=backend,
                              headline_timeout=0.0, collectives="torch", pg_timeout=600.0)
    return bench.launch(args, ["--gpus", "2", "--config", "c2", f"--backend={backend}"])


def test_launcher_runs_strong_shape_in_a_second_fresh_rank_set(tmp_path, monkeypatch, capsys):
    assert _fake_launch(tmp_path, monkeypatch, "ok") == 0
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] == 1.0 and out["strong_scaling"]["value"] == 2.0 and out["strong_scaling"]["num_envs_total"] == 256


def test_launcher_keeps_the_headline_when_the_strong_phase_fails_or_hangs(tmp_path, monkeypatch, capsys):
    for mode in ("fail", "hang"):
        assert _fake_launch(tmp_path, monkeypatch, mode) == 0
        out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert out["value"] == 1.0 and "error" in out["strong_scaling"]


def test_launcher_times_out_a_hung_headline_phase(tmp_path, monkeypatch, capsys):
    """A rank stuck in a collective must not hold the launcher forever: the headline rank set has a limit too; on expiry
    the children are terminated and the launcher exits non-zero without a JSON line."""
    rc = _fake_launch(tmp_path, monkeypatch, "headline-hang", cap=3)
    cap = capsys.readouterr()
    assert rc == 124 and cap.out.strip() == "" and "timed out" in cap.err


def test_shared_gpu_queue_cap_follows_the_parsed_backend(tmp_path, monkeypatch, capsys):
    """`--backend=gloo` (one token) is a shared-GPU rehearsal exactly like `--backend gloo`: the rank processes get
    GPU_MAX_HW_QUEUES=2; with nccl (one process per GPU) they do not."""
    for backend, want in (("gloo", "2"), ("nccl", "unset")):
        out = tmp_path / f"env_{backend}.txt"
        monkeypatch.setenv("FAKE_ENV_OUT", str(out))
        monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
        assert _fake_launch(tmp_path, monkeypatch, "ok", backend=backend) == 0
        capsys.readouterr()
        assert out.read_text() == want
