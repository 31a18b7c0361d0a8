"""The random-configuration soak of the batched nprobe path (scripts/fuzz_prescan.py): every drawn configuration runs
through the matrix-core list scan (fp16 shadow rows, the default) AND through the ordered-chain scan on the same handle
state in separate processes; all queries of both must agree bit for bit, and a sample must equal the C oracle.

In the suite it draws for VERS_FUZZ_SECONDS (default 25 s: four to five configurations, a different slice per first seed);
the long soak is the same test with VERS_FUZZ_SECONDS=600 / 1200 (577 + 1,134 + 566 + 852 configurations clean on the round-2 kernels)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_soak():
    budget = float(os.environ.get("VERS_FUZZ_SECONDS", "25"))
    seed = os.environ.get("VERS_FUZZ_SEED", "9100")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_prescan.py"), str(budget), seed], capture_output=True, text=True,
                       cwd=ROOT, timeout=budget * 3 + 600)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and "FAILED" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "random configurations" in r.stdout, r.stdout[-500:]


def test_random_single_queries_soak():
    """scripts/fuzz_single.py: single queries (host-pointer call and device pointers: coarse1_kernel -> the single query's list scans, single.hip.h -> merge or exact finish) against
    the same queries in a batch and an oracle sample, random shapes / metrics / modes.  15 s in the suite (≈ 200 configurations,
    40 k single queries); 2,100 configurations / 400,512 single queries clean in the round-3 soak."""
    budget = float(os.environ.get("VERS_FUZZ_SINGLE_SECONDS", "15"))
    seed = os.environ.get("VERS_FUZZ_SEED", "9100")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_single.py"), str(budget), seed], capture_output=True, text=True,
                       cwd=ROOT, timeout=budget * 3 + 600)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout + r.stderr, r.stdout[-3000:] + r.stderr[-2000:]
    assert "random configurations" in r.stdout, r.stdout[-500:]


def test_random_flat_queries_soak():
    """scripts/fuzz_flat.py: the flat index's single queries on its fp16 shadow against the f32 ordered chains and an oracle sample,
    random corpora (tile remainders, padding columns, ties, both metrics).  10 s in the suite (≈ 300 configurations); 8,664
    configurations / 112 k queries clean in the round-5 soak."""
    budget = float(os.environ.get("VERS_FUZZ_FLAT_SECONDS", "10"))
    seed = os.environ.get("VERS_FUZZ_SEED", "9100")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_flat.py"), str(budget), seed], capture_output=True, text=True,
                       cwd=ROOT, timeout=budget * 3 + 600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "random configurations" in r.stdout, r.stdout[-500:]
