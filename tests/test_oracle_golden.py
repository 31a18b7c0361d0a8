"""Both restatements against the committed fixtures (tests/golden/*.npz), bit for bit.

The fixtures were produced by tests/golden/make_golden.py only after the C and NumPy oracles
agreed; this test keeps them agreeing and detects drift in tests/datagen.py (input CRCs)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no
from tests.golden import make_golden as mg

ORACLES = [pytest.param(co, id="c"), pytest.param(no, id="numpy")]


def test_oracle_is_built_without_fma_or_fast_math():
    mk = open(os.path.join(os.path.dirname(co.__file__), "Makefile")).read()
    assert "-ffp-contract=off" in mk and "-fno-fast-math" in mk
    so = co.build()
    dis = subprocess.run(["objdump", "-d", so], capture_output=True, text=True).stdout
    assert "vfmadd" not in dis and "vfnmadd" not in dis


@pytest.mark.parametrize("o", ORACLES)
@pytest.mark.parametrize("cs", mg.FLAT_CASES, ids=lambda c: c["name"])
def test_flat_golden(o, cs, golden_flat):
    X = mg.corpus(cs); Q = mg.queries(cs["seed"] + 1, 3, cs["d"], X)
    assert list(golden_flat[cs["name"] + "/crc"]) == [mg.crc(X), mg.crc(Q)]
    for metric in (0, 1):
        for top_k in (1, 10, 64):
            ids = golden_flat[f"{cs['name']}/m{metric}/k{top_k}/ids"]
            db = golden_flat[f"{cs['name']}/m{metric}/k{top_k}/dist_bits"]
            for qi, q in enumerate(Q):
                i, d = o.search_exhaustive(X, q, top_k, metric)
                assert np.array_equal(i, ids[qi]) and np.array_equal(mg.bits(d), db[qi])


@pytest.mark.parametrize("o", ORACLES)
@pytest.mark.parametrize("cs", mg.KMEANS_CASES, ids=lambda c: c["name"])
def test_kmeans_and_search_golden(o, cs, golden_km):
    g, nm = golden_km, cs["name"]
    X = mg.corpus(cs); k, n, d = cs["k"], cs["n"], cs["d"]
    assert g[nm + "/crc"][0] == mg.crc(X)
    init = mg.init_draws(cs["seed"] ^ 0xABCD, cs["attempts"], k, n)
    assert np.array_equal(init, g[nm + "/init"])
    C0 = X[init[:k].astype(np.int64)]
    a0 = o.assign_to_clusters(X, C0)
    assert np.array_equal(a0, g[nm + "/assign0"])
    assert np.array_equal(mg.bits(o.update_centroids(X, a0, k)), g[nm + "/update0_bits"])
    assert mg.bits(np.array([o.kmeans_cost(X, C0, a0)]))[0] == g[nm + "/cost0_bits"][0]
    b = o.build_index(X, k, cs["attempts"], cs["iters"], init)
    assert np.array_equal(mg.bits(b["centroids"]), g[nm + "/build_C_bits"])
    assert np.array_equal(b["assignments"], g[nm + "/build_assign"])
    assert mg.bits(np.array([b["cost"]]))[0] == g[nm + "/build_cost_bits"][0]
    # invariants that follow from the source
    seen = np.concatenate([np.asarray(l, dtype=np.uint64) for l in b["ids"]])
    assert np.array_equal(np.sort(seen), np.arange(n, dtype=np.uint64))
    for c, l in enumerate(b["ids"]):
        assert np.all(np.diff(np.asarray(l, dtype=np.int64)) > 0) and np.all(b["assignments"][np.asarray(l, dtype=np.int64)] == c)
    # search after 3 adds
    values = X.copy(); ids = [list(l) for l in b["ids"]]; nassign = n
    extra = mg.dg.dist_u(cs["seed"] + 7, 3, d)
    for x, want in zip(extra, g[nm + "/add_clusters"]):
        c = o.add_cluster(b["centroids"], x)
        assert c == want
        ids[c].append(nassign); nassign += 1
        values = np.concatenate([values, x[None]], axis=0)
    Q = mg.queries(cs["seed"] + 3, 6, d, values); Q[1] = extra[1]
    assert list(g[nm + "/crc_q"]) == [mg.crc(Q), mg.crc(extra)]
    for top_k in (1, 10, 50):
        for tag, fn in [("search", lambda q: o.search_approximate(values, b["centroids"], ids, q, top_k))] + [
                (f"nprobe{p}", (lambda p: lambda q: o.search_nprobe(values, b["centroids"], ids, q, top_k, p))(p)) for p in (1, 4, k)]:
            gi = g[f"{nm}/{tag}/k{top_k}/ids"]; gd = g[f"{nm}/{tag}/k{top_k}/dist_bits"]; gc = g[f"{nm}/{tag}/k{top_k}/count"]
            for qi, q in enumerate(Q):
                i, dd = fn(q)
                assert len(i) == gc[qi]
                assert np.array_equal(i, gi[qi][:len(i)]) and np.array_equal(mg.bits(dd), gd[qi][:len(i)])
    # self-retrieval: a bit-identical stored row comes back first at distance exactly 0.0
    i, dd = o.search_approximate(values, b["centroids"], ids, values[n + 1], 1)
    assert i[0] == n + 1 and dd[0] == 0.0


def cos_case_state(o, cs, g):
    """build + 3 adds for one COS_CASES entry -> (X, values, centroids, ids, Q)"""
    nm = cs["name"]
    X = mg.corpus(cs); k, n, d = cs["k"], cs["n"], cs["d"]
    assert g[nm + "/crc"][0] == mg.crc(X)
    init = mg.init_draws(cs["seed"] ^ 0xABCD, cs["attempts"], k, n)
    assert np.array_equal(init, g[nm + "/init"])
    b = o.build_index(X, k, cs["attempts"], cs["iters"], init, metric=1)
    values = X.copy(); ids = [list(l) for l in b["ids"]]; nassign = n
    extra = mg.dg.dist_u(cs["seed"] + 7, 3, d)
    for x, want in zip(extra, g[nm + "/add_clusters"]):
        c = o.add_cluster(b["centroids"], x, metric=1)
        assert c == want
        ids[c].append(nassign); nassign += 1
        values = np.concatenate([values, x[None]], axis=0)
    Q = mg.queries(cs["seed"] + 3, 6, d, values); Q[1] = extra[1]
    assert list(g[nm + "/crc_q"]) == [mg.crc(Q), mg.crc(extra)]
    return X, b, values, ids, Q


@pytest.fixture(scope="session")
def golden_cos():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ivf_cosdist.npz"))


@pytest.mark.parametrize("o", ORACLES)
@pytest.mark.parametrize("cs", mg.COS_CASES, ids=lambda c: c["name"])
def test_cosine_distance_ivf_golden(o, cs, golden_cos):
    """the metric extension (1 - dot wherever ivfflat.rs calls squared_euclidean): both restatements == fixtures"""
    g, nm, k = golden_cos, cs["name"], cs["k"]
    X, b, values, ids, Q = cos_case_state(o, cs, g)
    assert np.array_equal(mg.bits(b["centroids"]), g[nm + "/build_C_bits"])
    assert np.array_equal(b["assignments"], g[nm + "/build_assign"])
    assert mg.bits(np.array([b["cost"]]))[0] == g[nm + "/build_cost_bits"][0]
    for top_k in (1, 10, 40):
        for nprobe in (0, 1, 4, k):
            gi = g[f"{nm}/nprobe{nprobe}/k{top_k}/ids"]; gd = g[f"{nm}/nprobe{nprobe}/k{top_k}/dist_bits"]; gc = g[f"{nm}/nprobe{nprobe}/k{top_k}/count"]
            for qi, q in enumerate(Q):
                i, dd = (o.search_approximate(values, b["centroids"], ids, q, top_k, metric=1) if nprobe == 0 else
                         o.search_nprobe(values, b["centroids"], ids, q, top_k, nprobe, metric=1))
                assert len(i) == gc[qi]
                assert np.array_equal(i, gi[qi][:len(i)]) and np.array_equal(mg.bits(dd), gd[qi][:len(i)])


def test_cosine_distance_micro_cases():
    """hand-checkable: 1 - dot in f32, sequential dot; ranking differs from L2 when centroids are not unit length"""
    C = np.array([[2.0, 0.0], [0.0, 1.0], [0.6, 0.6]], dtype=np.float32)
    q = np.array([1.0, 0.0], dtype=np.float32)
    for o in (co, no):
        # cos dist: 1-2 = -1, 1-0 = 1, 1-0.6 = 0.4 -> cluster 0 ; L2: 1, 2, 0.16+0.36=0.52 -> cluster 2
        assert o.add_cluster(C, q, metric=1) == 0 and o.add_cluster(C, q, metric=0) == 2
        vals = np.array([[1.0, 0.0], [0.5, 0.5], [3.0, 0.0], [0.0, 2.0]], dtype=np.float32)
        ids = [[2], [3], [0, 1]]
        i, d = o.search_approximate(vals, C, ids, q, 3, metric=1)
        # nearest list by cos dist is 0 -> id 2 (1-3 = -2); spill to list 2 (0.4): ids 0 (1-1 = 0), 1 (1-0.5 = 0.5)
        assert list(i) == [2, 0, 1] and list(d) == [-2.0, 0.0, 0.5]
        i, d = o.search_nprobe(vals, C, ids, q, 4, 3, metric=1)
        assert list(i) == [2, 0, 1, 3] and list(d) == [-2.0, 0.0, 0.5, 1.0]
