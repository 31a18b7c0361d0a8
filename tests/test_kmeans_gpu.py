"""GPU parity of the k-means steps (assign / update / cost) against the golden fixtures and the
oracle -- bit-exact: identical assignments, identical centroid and cost bits."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("cs", mg.KMEANS_CASES, ids=lambda c: c["name"])
def test_assign_update_cost_golden(cs, golden_km):
    g, nm = golden_km, cs["name"]
    X = mg.corpus(cs); k = cs["k"]
    init = g[nm + "/init"]
    C0 = X[init[:k].astype(np.int64)].copy()
    a, md = capi.kmeans_assign(X, C0, want_min_dist=True)
    assert np.array_equal(a, g[nm + "/assign0"])
    # the minimum distance is exactly D(x_i, c_{a_i}) in reference arithmetic
    want = np.array([co.squared_euclidean(X[i], C0[int(a[i])]) for i in range(0, X.shape[0], 37)], dtype=np.float32)
    assert np.array_equal(bits(md[::37]), bits(want))
    u = capi.kmeans_update(X, a, k)
    assert np.array_equal(bits(u), g[nm + "/update0_bits"])
    c = capi.kmeans_cost(X, C0, a)
    assert bits(np.array([c]))[0] == g[nm + "/cost0_bits"][0]


@pytest.mark.parametrize("n,d,k", [(1000, 4, 1), (777, 20, 3), (4096, 64, 130), (300, 300, 64), (50, 768, 65), (9, 5, 20)])
def test_assign_vs_oracle_shapes(n, d, k):
    X = dg.dist_u(11 + n, n, d); Cn = dg.dist_u(13 + k, k, d)
    if k >= 3:
        Cn[2] = Cn[0]  # duplicate centroid: ties must go to the lower index
    a = capi.kmeans_assign(X, Cn)
    assert np.array_equal(a, co.assign_to_clusters(X, Cn))
    u = capi.kmeans_update(X, a, k)
    assert np.array_equal(bits(u), bits(co.update_centroids(X, a, k)))
    assert bits(np.array([capi.kmeans_cost(X, Cn, a)]))[0] == bits(np.array([co.kmeans_cost(X, Cn, a)]))[0]


def test_assign_error_semantics():
    X = dg.dist_u(5, 10, 8)
    with pytest.raises(capi.VersError) as e:
        capi.kmeans_assign(X, np.zeros((0, 8), dtype=np.float32))
    assert e.value.status == capi.ERR_EMPTY           # ivfflat.rs:42 unwrap on None
    Cn = dg.dist_u(6, 3, 8); Cn[1, 2] = np.nan
    with pytest.raises(capi.VersError) as e:
        capi.kmeans_assign(X, Cn)
    assert e.value.status == capi.ERR_NAN             # ivfflat.rs:40 partial_cmp().unwrap()
    a = capi.kmeans_assign(X, Cn[1:2])                # a single centroid is never compared: no panic
    assert np.array_equal(a, np.zeros(10, dtype=np.uint64))
    assert capi.kmeans_assign(X[:0], Cn).shape == (0,)
