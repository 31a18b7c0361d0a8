"""Batched coarse quantiser on the f32 matrix cores (csrc/gemm.hip.h): MFMA pre-selection + exact re-score +
certificate must give the SAME probe lists -- hence bit-identical search results -- as the exact path, and the
certificate's fallback (forced here through the option coarse=2) must too."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd.index import IVFFlatIndex
n, d, k = 9000, 96, 300
X = dg.dist_c(0x61, n, d, 900, dg.default_sigma(d))
ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(3, 1, k, n))
Q = dg.dist_c(0x62, 150, d, 900, dg.default_sigma(d)); Q[5] = X[77]
for nprobe, top_k in [(8, 10), (0, 10), (40, 64)]:
    ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
    for qi in range(0, 150, 11):
        oi, od = (co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k) if nprobe == 0 else
                  co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe))
        assert cnt[qi] == len(oi), (nprobe, qi)
        assert np.array_equal(ids[qi, :len(oi)], oi), (nprobe, qi)
        assert np.array_equal(dist[qi, :len(oi)].view(np.uint32), od.view(np.uint32)), (nprobe, qi)
st = ix.coarse_stats()
print("STATS", st["mfma_batches"], st["fallback_queries"])
def check(ix, Q, top_k, nprobe, step):
    ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
    for qi in range(0, Q.shape[0], step):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe)
        assert cnt[qi] == len(oi) and np.array_equal(ids[qi, :len(oi)], oi), (nprobe, qi)
        assert np.array_equal(dist[qi, :len(oi)].view(np.uint32), od.view(np.uint32)), (nprobe, qi)
# uniform data: the centroids of a query are all about equally far, (nearly) every one of the P + 16 = 60 selected candidates
# can still reach the top P -- the exact re-score stages its rows in TWO passes (40 + 20: gemm.hip.h, kSelRows)
Xu = dg.dist_u(0x63, 8000, 64)
iu = IVFFlatIndex.build_index(120, 1, 2, Xu, init_indices=mg.init_draws(5, 1, 120, 8000))
check(iu, dg.dist_u(0x64, 96, 64), 10, 44, 7)
# more than 4096 centroids: the selection walks the row of G in chunks of 4096 values and merges the chunks' candidates
Xk = dg.dist_c(0x65, 24000, 32, 6000, dg.default_sigma(32))
ik = IVFFlatIndex.build_index(5000, 1, 1, Xk, init_indices=mg.init_draws(7, 1, 5000, 24000))
check(ik, dg.dist_c(0x66, 64, 32, 6000, dg.default_sigma(32)), 10, 12, 5)
print("EXTRA", iu.coarse_stats()["mfma_batches"], ik.coarse_stats()["mfma_batches"])
'''


def run(env_extra):
    env = dict(os.environ); env.update(env_extra); env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", BODY], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("STATS")][0].split()
    extra = [l for l in r.stdout.splitlines() if l.startswith("EXTRA")][0].split()
    if env_extra.get("VERS_OPTIONS") != "coarse=1":
        assert int(extra[1]) == 1 and int(extra[2]) == 1   # the uniform corpus and the 5000-centroid index ranked on the matrix cores too
    return int(line[1]), int(line[2])


def test_mfma_preselection_is_exact_and_usually_certified():
    batches, fallbacks = run({})
    assert batches == 3               # every batch of 150 queries went through the matrix cores
    assert fallbacks < 15             # the certificate passes for (nearly) all queries; failures are re-done exactly


def test_certificate_fallback_is_exact():
    batches, fallbacks = run({"VERS_OPTIONS": "coarse=2"})   # error bound forced to +inf: nothing certifies
    assert batches == 3 and fallbacks == 3 * 150


def test_exact_coarse_still_available():
    batches, _ = run({"VERS_OPTIONS": "coarse=1"})
    assert batches == 0
