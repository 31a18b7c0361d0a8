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
