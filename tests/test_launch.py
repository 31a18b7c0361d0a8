"""`python bench.py --gpus N` starts its own ranks (vers_amd/launch.py): the parent only spawns children with the
environment torch.distributed.run would give them, relays rank 0's stdout and fails when any child fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "launch_child.py")
PARENT = ("import sys; sys.path.insert(0, %r); from vers_amd.launch import spawn_ranks; "
          "sys.exit(spawn_ranks(%r, sys.argv[2:], int(sys.argv[1])))" % (ROOT, CHILD))


def run_parent(n, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", PARENT, str(n)] + list(extra), capture_output=True, text=True, timeout=300, env=env)


@pytest.mark.parametrize("n", [2, 3])
def test_spawn_ranks_relays_rank0_json(n):
    r = run_parent(n)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # ONE line on stdout: rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["sum"] == n * (n + 1) / 2 and out["local_rank"] == 0
    assert "noise from a non-zero rank" in r.stderr      # the other ranks' stdout is diverted


def test_spawn_ranks_fails_when_a_child_fails():
    r = run_parent(2, "--fail-rank", "1")
    assert r.returncode != 0
    assert "rank 1 exited with status 7" in r.stderr


def test_bench_parent_never_imports_torch_cuda():
    """the launcher module itself pulls in neither torch nor the HIP library"""
    code = ("import sys; sys.path.insert(0, %r); import vers_amd.launch; "
            "assert 'torch' not in sys.modules and 'vers_amd.capi' not in sys.modules" % ROOT)
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0
