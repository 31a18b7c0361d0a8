"""The metric extension end to end on the GPU: IVFFlat with cosine distance 1 - dot (base.rs:153-155)
in build_index, add, search_approximate and the nprobe extension -- against the committed fixtures
(tests/golden/ivf_cosdist.npz, produced by both oracles), against the oracle on larger batches that go through the three
matrix-core filters (coarse quantiser, list scan, k-means assign), and with every certificate forced to fail."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M1 = capi.METRIC_COSDIST


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def golden_cos():
    return np.load(os.path.join(ROOT, "tests", "golden", "ivf_cosdist.npz"))


@pytest.mark.parametrize("cs", mg.COS_CASES, ids=lambda c: c["name"])
def test_cosdist_golden(cs, golden_cos, tmp_path):
    g, nm = golden_cos, cs["name"]
    X = mg.corpus(cs); k, n, d = cs["k"], cs["n"], cs["d"]
    ix = IVFFlatIndex.build_index(k, cs["attempts"], cs["iters"], X, init_indices=g[nm + "/init"], metric=M1)
    assert np.array_equal(bits(ix.centroids), g[nm + "/build_C_bits"])
    assert np.array_equal(ix.assignments, g[nm + "/build_assign"])
    assert bits(np.array([ix.cost]))[0] == g[nm + "/build_cost_bits"][0]
    extra = dg.dist_u(cs["seed"] + 7, 3, d)
    for x, want in zip(extra, g[nm + "/add_clusters"]):
        c, _ = ix.add(x)
        assert c == want
    # save -> load keeps working with the metric given again (the file has no metric field)
    p = os.path.join(tmp_path, "cos.index")
    ix.save_index(p)
    re = IVFFlatIndex.load_index(p, d, metric=M1)
    Q = mg.queries(cs["seed"] + 3, 6, d, ix.values); Q[1] = extra[1]
    for index in (ix, re):
        for top_k in (1, 10, 40):
            for nprobe in (0, 1, 4, k):
                ids, dist, cnt = index.search_batch(Q, top_k, nprobe)
                gi = g[f"{nm}/nprobe{nprobe}/k{top_k}/ids"]; gd = g[f"{nm}/nprobe{nprobe}/k{top_k}/dist_bits"]; gc = g[f"{nm}/nprobe{nprobe}/k{top_k}/count"]
                assert np.array_equal(cnt, gc), (nprobe, top_k)
                for qi in range(Q.shape[0]):
                    c = int(gc[qi])
                    assert np.array_equal(ids[qi, :c], gi[qi, :c]) and np.array_equal(bits(dist[qi, :c]), gd[qi, :c]), (nprobe, top_k, qi)
        # one query at a time = Index::search_approximate
        r = index.search_approximate(Q[2], 10)
        assert [i for i, _ in r] == list(g[f"{nm}/nprobe0/k10/ids"][2][:len(r)])
    ix.close(); re.close()


BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
M1 = capi.METRIC_COSDIST
total = 0
for seed, n, d, k, b, nprobe, top_ks, scale in [(0xC1, 9000, 96, 48, 160, 8, (1, 10, 30, 100), False), (0xC2, 4000, 300, 32, 96, 6, (10,), True),
                                                (0xC3, 3000, 768, 24, 70, 5, (10,), False)]:
    X = dg.dist_c(seed, n, d, 4 * k, dg.default_sigma(d))
    if scale:  # rows of different lengths: cosine distance 1 - dot is then NOT a monotone function of the L2 distance
        X = (X * (0.5 + (np.arange(n) % 5)[:, None] * 0.375)).astype(np.float32)
    init = mg.init_draws(seed, 1, k, n)
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=init, metric=M1)
    o = co.build_index(X, k, 1, 3, init, metric=1)
    assert np.array_equal(ix.assignments, o["assignments"]), "assignments"
    assert np.array_equal(ix.centroids.view(np.uint32), o["centroids"].view(np.uint32)), "centroids"
    assert np.float32(ix.cost).view(np.uint32) == np.float32(o["cost"]).view(np.uint32), "cost"
    Q = dg.dist_c(seed + 0x100, b, d, 4 * k, dg.default_sigma(d)); Q[3] = X[17]
    for top_k in top_ks:
        for np_ in (0, nprobe):
            ids, dist, cnt = ix.search_batch(Q, top_k, np_)
            for qi in range(0, b, 7):
                oi, od = (co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k, metric=1) if np_ == 0 else
                          co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, np_, metric=1))
                assert cnt[qi] == len(oi), (np_, top_k, qi)
                assert np.array_equal(ids[qi, :len(oi)], oi), (np_, top_k, qi, ids[qi, :len(oi)], oi)
                assert np.array_equal(dist[qi, :len(oi)].view(np.uint32), od.view(np.uint32)), (np_, top_k, qi)
            total += 1
    st = ix.prescan_stats(); cs = ix.coarse_stats()
    print("stats", seed, st, cs, capi.assign_stats())
    assert st["batches"] >= 1 and cs["mfma_batches"] >= 1   # the batches went through the matrix-core filters
    if FORCED:
        assert st["fallback_queries"] >= b and cs["fallback_queries"] >= b   # ... and every certificate failed
    ix.close()
print("checked", total)
'''


def run_body(env_extra, forced):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", f"FORCED = {forced}\n" + BODY], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "checked" in r.stdout


def test_cosdist_matrix_core_paths_match_oracle():
    run_body({"VERS_OPTIONS": "assign=2"}, False)          # k-means assign through the matrix cores too


def test_cosdist_with_every_certificate_forced_to_fail():
    run_body({"VERS_OPTIONS": "prescan=2,coarse=2,assign=2"}, True)
