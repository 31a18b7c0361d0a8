"""The list scan's certificate, audited with measurements.

vers_ivf_test_last_vals hands back the raw pre-filter values the matrix-core list scan produced for a query -- `val` =
|x|^2 - 2 <x~, q> (or -<x~, q>) exactly as the certificate saw it -- with the bound the certificate charges each candidate.
For every dumped (row, val) the reference's ordered-chain distance is recomputed on the host and
        | val + |q|^2 - D_ref |  /  bound   (cosine: | 1 + val - D_ref | / bound)
must be <= 1; the worst ratio per configuration is printed (DESIGN.md section 1 quotes them).  Then the adversarial cases:
rows whose fp16 rounding errors all point along the query (the Cauchy-Schwarz step of the shadow term is attained), and more
near-ties inside the window than a candidate list holds (the certificate must FAIL and the exact re-scan decide)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi, testhooks
from vers_amd.index import IVFFlatIndex

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def d_ref(x, q, metric):
    """the reference's ordered f32 chain (base.rs:119-126; cosine distance: 1 - the sequential dot, base.rs:91-93,153-155)"""
    return float(np.float32(1.0) - co.dot(x, q)) if metric else float(co.squared_euclidean(x, q))


def worst_ratio(ix, X_all, Q, metric, top_k, nprobe, queries):
    ix.search_batch(Q, top_k, nprobe)
    worst, n_vals = 0.0, 0
    for qi in queries:
        ids, vals, bnd, info = testhooks.last_vals(ix, qi)
        assert len(ids) > 0 and info["metric"] == metric
        for vid, v, b in zip(ids, vals, bnd):
            D = d_ref(X_all[int(vid)], Q[qi], metric)
            err = abs((1.0 + float(v) - D) if metric else (float(v) + info["qn"] - D))
            assert np.isfinite(b) and b > 0
            worst = max(worst, err / b)
            n_vals += 1
    return worst, n_vals


def corpus(kind, n, d, seed):
    if kind == "dist_c":
        return dg.dist_c(seed, n, d, 48, dg.default_sigma(d))
    if kind == "dist_u":
        return dg.dist_u(seed, n, d)
    if kind == "norm_300":
        return (dg.dist_c(seed, n, d, 48, dg.default_sigma(d)) * np.float32(300.0)).astype(np.float32)
    if kind == "mixed_subnormal":     # a third of the elements in fp16's SUBNORMAL range (below 2^-14), the rest ordinary
        X = dg.dist_c(seed, n, d, 48, dg.default_sigma(d))
        X[:, ::3] *= np.float32(2.0 ** -13)
        return X
    raise ValueError(kind)


@pytest.mark.parametrize("shadow", [1, 0])
@pytest.mark.parametrize("metric", [0, 1])
def test_every_dumped_val_is_inside_its_bound(metric, shadow):
    n, d, k, b, top_k, nprobe = 6000, 96, 12, 64, 30, 6
    capi.set_option("shadow", shadow)
    try:
        for kind in ("dist_c", "dist_u", "norm_300", "mixed_subnormal"):
            X = corpus(kind, n, d, 0x900 + metric)
            Q = corpus(kind, b, d, 0x910 + metric)
            ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x900, 1, k, n), metric=metric)
            assert bool(ix.shadow_state()["active"]) == bool(shadow)
            w, nv = worst_ratio(ix, X, Q, metric, top_k, nprobe, range(0, b, 5))
            print(f"metric {metric} {'fp16 shadow' if shadow else 'f32 rows   '} {kind:16s}: worst |val - exact| / bound = {w:.4f} over {nv} dumped vals")
            assert w <= 1.0, (kind, w)
            assert ix.prescan_stats()["batches"] >= 1
            if shadow:   # the WIDE candidate lists (results of 49 .. 200 keys: four keys per lane, hi-only query blocks) dump their vals the same way
                b0 = ix.prescan_stats()["batches"]
                w, nv = worst_ratio(ix, X, Q, metric, 100, nprobe, range(0, b, 9))
                print(f"metric {metric} fp16 shadow, wide lists {kind:16s}: worst |val - exact| / bound = {w:.4f} over {nv} dumped vals")
                assert w <= 1.0 and ix.prescan_stats()["batches"] == b0 + 1, (kind, w)
            ix.close()
    finally:
        capi.set_option("shadow", 1)


def test_shadow_rounding_errors_aligned_with_the_query():
    """Worst case of the shadow term 2 R |q|: every element of x sits 0.49 ulp(fp16) off its fp16 value, on the side of the
    query's sign, all elements in one binade -- the residual x - fp16(x) is parallel to q and Cauchy-Schwarz is attained."""
    n, d, k, b, top_k = 4096, 128, 8, 32, 10
    rng = np.random.default_rng(0x5AD)
    h = (0.03125 + rng.integers(0, 1024, (n, d)) * 2.0 ** -15).astype(np.float32)        # fp16 values in [2^-5, 2^-4): ulp 2^-15
    sgn = rng.choice([-1.0, 1.0], d).astype(np.float32)
    X = (h * rng.choice([-1.0, 1.0], (n, d))).astype(np.float32)
    X = (X + np.sign(X) * 0 + (sgn[None, :] * np.float32(0.49 * 2.0 ** -15))).astype(np.float32)   # + 0.49 ulp along sgn: fp16(x) is the unshifted value
    assert np.array_equal(X.astype(np.float16).astype(np.float32) != X, np.ones_like(X, dtype=bool))
    Q = np.tile((sgn / np.sqrt(np.float32(d)))[None, :], (b, 1)).astype(np.float32)
    Q += (1e-3 * rng.standard_normal((b, d))).astype(np.float32)                           # distinct queries, still along sgn
    ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x5AD, 1, k, n))
    assert ix.shadow_state()["active"]
    w, nv = worst_ratio(ix, X, Q, 0, top_k, k, range(0, b, 3))
    print(f"aligned shadow residuals: worst |val - exact| / bound = {w:.4f} over {nv} vals (the Cauchy-Schwarz step is attained: this is as tight as the bound gets)")
    assert 0.3 < w <= 1.0, w
    ids, dist, cnt = ix.search_batch(Q, top_k, k)
    for qi in range(0, b, 5):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, k)
        assert np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od))
    ix.close()


def test_more_near_ties_than_a_candidate_list_holds():
    """70 rows within a few f32 ulps of each other around the query's nearest neighbour: all of them are inside the window,
    the list (top_k + 24 keys) is full of them and cannot certify -- the query must go to the exact re-scan and come out
    with the reference's order (ties by list position)."""
    n, d, k, b, top_k = 5000, 64, 10, 32, 10
    X = dg.dist_c(0x71E, n, d, 40, dg.default_sigma(d))
    base = X[17].copy()
    for t in range(70):
        r = base.copy()
        r[t % d] = np.nextafter(r[t % d], np.float32(2.0) if t % 2 else np.float32(-2.0))   # one element, one ulp
        X[200 + 13 * t] = r
    Q = dg.dist_c(0x71F, b, d, 40, dg.default_sigma(d))
    Q[0] = base; Q[1] = base * np.float32(1.0001)
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x71E, 1, k, n))
    before = ix.prescan_stats()["fallback_queries"]
    ids, dist, cnt = ix.search_batch(Q, top_k, 4)
    st = ix.prescan_stats()
    assert st["batches"] >= 1 and st["fallback_queries"] - before >= 2      # the two queries at the cluster of near-ties
    for qi in range(b):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, 4)
        assert cnt[qi] == len(oi) and np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od)), qi
    ix.close()


def test_hi_only_query_block_every_val_inside_its_bound_and_query_residuals_aligned_with_the_rows():
    """d = 1536: the query block of the shadow scan is fp16 hi ONLY (round 5) and the certificate charges the query's measured
    residual |q' - fp16(q')| (|x| + R).  Ordinary corpora first; then the adversarial one for THAT term: every element of the
    scaled query -2q sits 0.49 ulp(fp16) off its fp16 value on the side of the rows' common sign pattern, so that q' - fp16(q') is
    parallel to x~ and the Cauchy-Schwarz step is attained."""
    n, d, k, b, top_k, nprobe = 2500, 1536, 8, 32, 10, 4
    for kind in ("dist_c", "mixed_subnormal"):
        X = corpus(kind, n, d, 0x920); Q = corpus(kind, b, d, 0x921)
        ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x920, 1, k, n))
        w, nv = worst_ratio(ix, X, Q, 0, top_k, nprobe, range(0, b, 8))
        assert testhooks.last_vals(ix, 0)[3]["shadow"] == 2, "d = 1536 on the shadow should run hi-only query blocks"
        print(f"hi-only query block, {kind:16s}: worst |val - exact| / bound = {w:.4f} over {nv} dumped vals")
        assert w <= 1.0, (kind, w)
        ix.close()
    rng = np.random.default_rng(0x5AE)
    sgn = rng.choice([-1.0, 1.0], d).astype(np.float32)
    # rows: exactly representable in fp16 (no shadow residual of their own), all along sgn, |x| ~ 1
    X = (sgn[None, :] * (0.015625 + rng.integers(0, 1024, (n, d)) * 2.0 ** -16)).astype(np.float32)      # fp16 values in [2^-6, 2^-5)
    assert np.array_equal(X.astype(np.float16).astype(np.float32), X)
    # queries: -2 q = -(fp16 value in [2^-5, 2^-4)) * sgn - 0.49 ulp * sgn  ->  the residual of q' is -0.49 ulp * sgn: parallel to every row
    hq = (0.03125 + rng.integers(0, 1024, (b, d)) * 2.0 ** -15).astype(np.float32)
    Qp = (-(hq + np.float32(0.49 * 2.0 ** -15)) * sgn[None, :]).astype(np.float32)
    Q = (Qp / np.float32(-2.0)).astype(np.float32)
    assert np.array_equal((np.float32(-2.0) * Q), Qp) and not np.any((Qp.astype(np.float16).astype(np.float32)) == Qp)
    ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x5AE, 1, k, n))
    w, nv = worst_ratio(ix, X, Q, 0, top_k, k, range(0, b, 6))
    print(f"aligned query residuals (hi-only block): worst |val - exact| / bound = {w:.4f} over {nv} vals")
    assert 0.3 < w <= 1.0, w
    ids, dist, cnt = ix.search_batch(Q, top_k, k)
    for qi in range(0, b, 5):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, k)
        assert np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od))
    ix.close()
