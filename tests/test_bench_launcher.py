"""Host tests of bench.py's launcher: `python bench.py --gpus N` starts N ranks itself (one process per GPU, the way the
reference's distributed tests start themselves, test/test_distributed_sea_ice.jl:41-54), relays rank 0's JSON line and
never falls back to fewer ranks.  No GPU needed: the plan is printed, the process plumbing runs over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_print_launch_is_one_rank_per_gpu_on_localhost(n):
    p = run("--gpus", str(n), "--steps", "3", "--print-launch")
    assert p.returncode == 0, p.stderr
    plan = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = plan["launch"]
    assert plan["ranks"] == n and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert f"--nproc-per-node={n}" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", str(n), "--steps", "3"]              # the ranks get the parent's arguments, without --print-launch
    assert plan["would_run"] == (plan["visible_gpus"] >= n)


def test_refuses_to_run_on_fewer_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    p = run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode == 2 and "refusing to fall back" in p.stderr and p.stdout.strip() == ""


def test_gpus_must_match_world_size():
    p = run("--gpus", "2", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
    p = run("--gpus", "3")
    assert p.returncode != 0


def test_parent_starts_ranks_and_relays_rank0_line():
    p = run("--gpus", "2", "--self-test-launch")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                         # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["self_test"] and out["launched_by_parent"]
