"""GPU parity of the brute-force scan (vers_flat_*) against the oracle and the golden fixtures.
Bar: ids identical and in identical order, distances BIT-identical (the kernel keeps the
reference's sequential f32 summation order, so no tolerance is needed)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("cs", mg.FLAT_CASES, ids=lambda c: c["name"])
def test_flat_golden_single_and_batched(cs, golden_flat):
    X = mg.corpus(cs); Q = mg.queries(cs["seed"] + 1, 3, cs["d"], X)
    fc = capi.FlatCorpus(cs["d"]); fc.upload(X)
    for metric in (0, 1):
        for top_k in (1, 10, 64):
            gi = golden_flat[f"{cs['name']}/m{metric}/k{top_k}/ids"]
            gd = golden_flat[f"{cs['name']}/m{metric}/k{top_k}/dist_bits"]
            ids, dist, cnt = fc.search(Q, top_k, metric)            # batched path (QG=8)
            assert list(cnt) == [min(top_k, cs["n"])] * 3
            assert np.array_equal(ids, gi) and np.array_equal(bits(dist), gd)
            for qi in range(3):                                       # single-query path (QG=1)
                i1, d1, c1 = fc.search(Q[qi], top_k, metric)
                assert np.array_equal(i1[0], gi[qi]) and np.array_equal(bits(d1[0]), gd[qi])
    fc.close()


@pytest.mark.parametrize("n,d,b,k", [(5000, 128, 1, 10), (5000, 128, 19, 10), (3001, 300, 9, 7), (257, 768, 8, 64),
                                     (64, 4, 3, 5), (65, 5, 2, 64), (1, 16, 1, 1), (100000, 32, 2, 10)])
def test_flat_vs_oracle_random(n, d, b, k):
    X = dg.dist_u(0xABC0 + n, n, d); Q = dg.dist_u(0xDEF0 + n, b, d)
    fc = capi.FlatCorpus(d); fc.upload(X)
    ids, dist, cnt = fc.search(Q, k)
    for qi in range(b):
        oi, od = co.search_exhaustive(X, Q[qi], k)
        assert cnt[qi] == len(oi)
        assert np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od))
    fc.close()


def test_flat_edge_cases():
    d = 12
    X = dg.dist_u(7, 50, d)
    fc = capi.FlatCorpus(d)
    # empty corpus -> zero results
    fc.upload(X[:0])
    ids, dist, cnt = fc.search(X[0], 5)
    assert cnt[0] == 0
    fc.upload(X)
    # top_k == 0 -> empty
    ids, dist, cnt = fc.search(X[:2], 0)
    assert list(cnt) == [0, 0]
    # top_k > n -> n results, all rows, self first at exactly 0.0
    ids, dist, cnt = fc.search(X[3], 64)
    assert cnt[0] == 50 and ids[0, 0] == 3 and dist[0, 0] == 0.0
    assert sorted(ids[0, :50]) == list(range(50))
    # NaN -> the reference panics; here status VERS_ERR_NAN
    q = X[0].copy(); q[1] = np.nan
    with pytest.raises(capi.VersError) as e:
        fc.search(q, 3)
    assert e.value.status == capi.ERR_NAN
    ids, dist, cnt = fc.search(X[0], 3)   # the latch is cleared, the handle still works
    assert ids[0, 0] == 0
    # top_k beyond one key per lane: 64 ranks per pass, still min(top_k, n) results (utils.rs:79 take(k) has no cap)
    ids, dist, cnt = fc.search(X[0], 65)
    assert cnt[0] == 50 and sorted(ids[0, :50]) == list(range(50))
    fc.close()


def test_flat_results_wider_than_a_wave():
    """utils::search_exhaustive (utils.rs:68-82) takes ANY k: stable sort of all n distances, take(k).  top_k in {65, 256, 1000}
    comes 64 ranks per pass on the device; single query and batch, both metrics, against the oracle bit for bit."""
    n, d = 3000, 40
    X = dg.dist_c(0x41, n, d, 30, dg.default_sigma(d))
    X[100] = X[7]; X[2000] = X[7]                      # exact duplicates: ties are ordered by index across pass boundaries too
    Q = np.concatenate([dg.dist_c(0x42, 4, d, 30, dg.default_sigma(d)), X[7:8]])
    fc = capi.FlatCorpus(d)
    fc.upload(X)
    for metric in (0, 1):
        for top_k in (65, 256, 1000, 3500):
            ids, dist, cnt = fc.search(Q, top_k, metric)
            i1, d1, c1 = fc.search(Q[4], top_k, metric)
            for qi in range(Q.shape[0]):
                oi, od = co.search_exhaustive(X, Q[qi], top_k, metric)
                m = min(top_k, n)
                assert cnt[qi] == m and len(oi) == m
                assert np.array_equal(ids[qi, :m], oi) and np.array_equal(bits(dist[qi, :m]), bits(od)), (metric, top_k, qi)
            assert c1[0] == min(top_k, n) and np.array_equal(i1[0, :c1[0]], ids[4, :c1[0]]) and np.array_equal(bits(d1[0, :c1[0]]), bits(dist[4, :c1[0]]))
    fc.close()


def test_flat_host_pitch_of_rust_vector_layout():
    # Rust Vec<Vector<300>> has a 1280-byte pitch (#[repr(align(256))], base.rs:14-17)
    n, d = 500, 300
    X = dg.dist_u(99, n, d)
    padded = np.zeros((n, 320), dtype=np.float32); padded[:, :d] = X; padded[:, d:] = 123.0  # junk in the padding
    fc = capi.FlatCorpus(d)
    capi.check(capi.lib().vers_flat_upload(fc._h, padded.ctypes.data_as(capi._vp), n, 1280)); fc.n = n
    ids, dist, cnt = fc.search(X[10], 10)
    oi, od = co.search_exhaustive(X, X[10], 10)
    assert np.array_equal(ids[0], oi) and np.array_equal(bits(dist[0]), bits(od))
    fc.close()


SHADOW_BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from vers_amd import capi
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
checked = 0
for n, d, kind in [(40000, 128, "u"), (9000, 300, "c"), (130, 64, "u"), (70000, 48, "ties")]:
    X = dg.dist_u(0x51AD + n, n, d) if kind != "c" else dg.dist_c(0x51AD + n, n, d, 40, dg.default_sigma(d))
    if kind == "ties":
        X[n // 2:] = X[: n - n // 2]          # every row twice: equal distances, the lower index first (utils.rs:77)
    Q = dg.dist_u(0x0DD + n, 6, d); Q[2] = X[n // 3]
    fc = capi.FlatCorpus(d); fc.upload(X)
    for metric in (0, 1):
        for top_k in (1, 10, 48, 58, 64):     # (58, 64: beyond the shadow path's k + 16 keys of slack <= 64 -- the ordered chains)
            for single_shadow in (1, 0):
                capi.set_option("single_shadow", single_shadow)
                for qi in range(6):
                    oi, od = co.search_exhaustive(X, Q[qi], top_k, metric=metric)
                    ids, dist, cnt = fc.search(Q[qi], top_k, metric)
                    assert cnt[0] == len(oi), (n, d, metric, top_k, qi)
                    assert np.array_equal(ids[0, :len(oi)], oi) and np.array_equal(bits(dist[0, :len(oi)]), bits(od)), (n, d, kind, metric, top_k, single_shadow, qi)
                    checked += 1
    capi.set_option("single_shadow", 1)
    fc.close()
print("checked", checked)
'''


def _run_shadow_body(env_extra):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SHADOW_BODY], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "checked" in r.stdout


def test_flat_single_query_on_the_shadow_and_on_the_f32_rows():
    """one query streams the flat corpus' fp16 shadow (flat1h_kernel + the inverted lists' exact finish) or, with
    vers_set_option("single_shadow", 0), its f32 rows through the ordered chains: the oracle's bits either way"""
    _run_shadow_body({})


def test_flat_single_query_with_every_certificate_forced_to_fail():
    _run_shadow_body({"VERS_OPTIONS": "prescan=2"})


def test_flat_single_query_without_a_shadow():
    _run_shadow_body({"VERS_SHADOW": "0"})
