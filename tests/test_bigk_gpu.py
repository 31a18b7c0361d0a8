"""BASELINE.json cfg5's cluster count in the suite: build_index with k = 65536 clusters (small N and d so that the CPU
oracle finishes in seconds) through the matrix-core assign path, against the oracle bit for bit -- 512 column tiles of
the assign GEMM, duplicate initial centroids and ~3/4 empty clusters (k > N), the per-point selection spread over 16
waves, the certificate, and the exact re-scan of the uncertified points."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
n, d, k = 12000, 8, 65536
X = dg.dist_c(0xB16, n, d, 3000, dg.default_sigma(d))
init = (dg.mix64(np.uint64(0xB16) + np.arange(k, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)   # with replacement: many duplicates
ix = IVFFlatIndex.build_index(k, 1, 1, X, init_indices=init)
o = co.build_index(X, k, 1, 1, init)
assert np.array_equal(ix.assignments, o["assignments"])
assert np.array_equal(ix.centroids.view(np.uint32), o["centroids"].view(np.uint32))
assert np.float32(ix.cost).view(np.uint32) == np.float32(o["cost"]).view(np.uint32)
lens = ix.list_lengths()
assert int(lens.sum()) == n and int((lens == 0).sum()) > k // 2
pts, fb = capi.assign_stats()
print("assign through the matrix cores:", pts, "points,", fb, "re-done exactly")
assert pts >= 2 * n
Q = dg.dist_c(0xB17, 40, d, 3000, dg.default_sigma(d))
ids, dist, cnt = ix.search_batch(Q, 10, 32)
for qi in range(0, 40, 5):
    oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], 10, 32)
    assert cnt[qi] == len(oi) and np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(dist[qi, :len(oi)].view(np.uint32), od.view(np.uint32))
print("bigk ok")
'''


def test_build_index_k65536_matches_oracle():
    env = dict(os.environ); env["VERS_OPTIONS"] = "assign=2"
    r = subprocess.run([sys.executable, "-c", BODY], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0 and "bigk ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
