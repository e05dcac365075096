"""`python bench.py --gpus N` without a torch.distributed.run parent must start its own N ranks as a child
process before anything touches the GPU (the driver's scaling run may invoke it that way)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and "rank_check" in l]
    return p, lines


def test_bench_self_launches_its_ranks_before_any_gpu_call():
    p, lines = _run(None, "--gpus", "2", "--rank-check")
    assert p.returncode == 0, p.stderr[-2000:]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 and l["cuda_initialized"] is False for l in lines)


def test_bench_under_an_external_launcher_does_not_relaunch():
    # what the driver does: torch.distributed.run has already set WORLD_SIZE / RANK for this process
    p, lines = _run({"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, "--gpus", "2", "--rank-check")
    assert p.returncode == 0, p.stderr[-2000:]
    assert lines == [{"rank_check": True, "rank": 1, "world": 2, "local_rank": 1, "cuda_initialized": False}]


def test_bench_rejects_a_world_that_differs_from_gpus():
    p, _ = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--rank-check")
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)
