"""Randomised sweep of shapes through the C ABI against the oracle (bit-exact): odd dimensions, tiny and
single-row lists, more lists than vectors, batch sizes around the query-group widths (8/16) and the MFMA
threshold (32), every top_k/nprobe regime incl. results and probe counts wider than one key per lane, both
metrics, the batched list scan fed by the fp16 shadow rows (default) and by the f32 rows, adds between searches."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


CASES = [
    # n, d, k, iters, b, seeds
    (50, 1, 3, 2, 5), (200, 7, 40, 3, 17), (333, 33, 5, 2, 9), (1000, 65, 64, 2, 33), (1500, 130, 10, 3, 64),
    (64, 64, 64, 1, 16), (4097, 16, 2, 2, 31), (900, 300, 30, 2, 40), (2500, 20, 300, 2, 130),
]


@pytest.fixture
def shadow_option(request):
    capi.set_option("shadow", request.param)   # (the handle reads it at build time)
    yield request.param
    capi.set_option("shadow", 1)


@pytest.mark.parametrize("shadow_option", [1, 0], ids=["fp16shadow", "f32rows"], indirect=True)
@pytest.mark.parametrize("metric", [0, 1], ids=["l2sq", "cosdist"])
@pytest.mark.parametrize("n,d,k,iters,b", CASES)
def test_random_shapes(n, d, k, iters, b, metric, shadow_option):
    X = dg.dist_c(n + d, n, d, max(2, k // 2), dg.default_sigma(d))
    if metric:  # rows of different lengths: 1 - dot is then not a monotone function of the L2 distance
        X = (X * (0.5 + (np.arange(n) % 5)[:, None] * 0.375)).astype(np.float32)
    init = mg.init_draws(n ^ d, 1, k, n)
    ix = IVFFlatIndex.build_index(k, 1, iters, X, init_indices=init, metric=metric)
    assert ix.shadow_state()["active"] == bool(shadow_option)
    ob = co.build_index(X, k, 1, iters, init, metric=metric)
    assert np.array_equal(ix.assignments, ob["assignments"]) and np.array_equal(bits(ix.centroids), bits(ob["centroids"]))
    assert bits(np.array([ix.cost]))[0] == bits(np.array([ob["cost"]]))[0]
    for x in dg.dist_u(n + 5, 3, d):
        ix.add(x)
    Q = dg.dist_c(n + d + 1, b, d, max(2, k // 2), dg.default_sigma(d))
    Q[0] = ix.values[n // 2]
    total = ix.values.shape[0]
    for top_k in sorted({1, min(7, total), min(64, total), min(150, total)}):   # 150: results wider than one key per lane
        for nprobe in sorted({0, 1, min(3, k), min(k, 64), min(k, 100)}):          # 100: more ranked lists than one key per lane
            ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
            for qi in list(range(0, b, max(1, b // 6))) + [b - 1]:
                if nprobe == 0:
                    oi, od = co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k, metric=metric)
                else:
                    oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe, metric=metric)
                assert cnt[qi] == len(oi), (top_k, nprobe, qi)
                assert np.array_equal(ids[qi, :len(oi)], oi), (top_k, nprobe, qi)
                assert np.array_equal(bits(dist[qi, :len(oi)]), bits(od)), (top_k, nprobe, qi)
            # single-query path for one of them
            i1, d1, c1 = ix.search_batch(Q[b // 2], top_k, nprobe)
            assert c1[0] == cnt[b // 2] and np.array_equal(i1[0, :c1[0]], ids[b // 2, :c1[0]]) and np.array_equal(bits(d1[0, :c1[0]]), bits(dist[b // 2, :c1[0]]))
    # exhaustive == utils::search_exhaustive over the same values, both metrics, batched and single
    for m2 in (0, 1):
        ids, dist, cnt = ix.search_exhaustive(Q[:min(b, 9)], min(10, total), m2)
        for qi in range(min(b, 9)):
            oi, od = co.search_exhaustive(ix.values, Q[qi], min(10, total), m2)
            assert np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od)), (m2, qi)
    ix.close()
