"""Generates tests/golden/*.npz -- expected outputs of the vers IVFFlat hot path.

The reference has no golden vectors and cannot be built in this environment
(SURVEY.md section 8c), so the fixtures come from TWO independent restatements
of the Rust source -- oracle/vers_oracle.c (C) and oracle/np_oracle.py (NumPy
float32) -- which must agree bit for bit before anything is written.  Inputs
are not stored: they are regenerated from (seed, shape) by tests/datagen.py
(integer hashing + exact float ops only); a CRC of the input bytes is stored so
generator drift is detected.

Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import c_oracle as co  # noqa: E402
from oracle import np_oracle as no  # noqa: E402
from tests import datagen as dg  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == "f":
        ok = a.shape == b.shape and np.array_equal(bits(a), bits(b))
    else:
        ok = a.shape == b.shape and np.array_equal(a.astype(np.uint64), b.astype(np.uint64))
    if not ok:
        raise SystemExit(f"C and NumPy oracles disagree on {what}")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def corpus(spec):
    """spec = dict(kind 'u'|'c', seed, n, d, [dup])."""
    if spec["kind"] == "u":
        X = dg.dist_u(spec["seed"], spec["n"], spec["d"])
    else:
        X = dg.dist_c(spec["seed"], spec["n"], spec["d"], spec["n_modes"], dg.default_sigma(spec["d"]))
    if spec.get("dup"):
        X = X[np.arange(spec["n"]) % spec["dup"]].copy()  # exact duplicates -> distance ties
    return X


def queries(seed, nq, d, X):
    Q = dg.dist_u(seed, nq, d)
    Q[0] = X[X.shape[0] // 3]  # a corpus row: self distance is exactly 0.0
    return Q


FLAT_CASES = [
    dict(name="flat_n2000_d128", kind="u", seed=0x5EED0001, n=2000, d=128),
    dict(name="flat_n1500_d300", kind="c", seed=0x5EED0011, n=1500, d=300, n_modes=12),
    dict(name="flat_n777_d3_dup", kind="u", seed=0x5EED0021, n=777, d=3, dup=16),
    dict(name="flat_n1024_d768", kind="u", seed=0x5EED0031, n=1024, d=768),
    dict(name="flat_n130_d20_dup", kind="u", seed=0x5EED0041, n=130, d=20, dup=7),
]


def make_flat():
    out = {}
    for cs in FLAT_CASES:
        X = corpus(cs)
        Q = queries(cs["seed"] + 1, 3, cs["d"], X)
        for metric in (0, 1):
            for top_k in (1, 10, 64):
                ids_all, dist_all = [], []
                for q in Q:
                    ic, dc = co.search_exhaustive(X, q, top_k, metric)
                    i_n, dn = no.search_exhaustive(X, q, top_k, metric)
                    same(ic, i_n, f"{cs['name']} ids"); same(dc, dn, f"{cs['name']} dist")
                    ids_all.append(ic); dist_all.append(bits(dc))
                out[f"{cs['name']}/m{metric}/k{top_k}/ids"] = np.stack(ids_all)
                out[f"{cs['name']}/m{metric}/k{top_k}/dist_bits"] = np.stack(dist_all)
        out[f"{cs['name']}/crc"] = np.array([crc(X), crc(Q)], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "flat.npz"), **out)
    print("flat.npz", len(out), "arrays")


KMEANS_CASES = [
    dict(name="km_n3000_d32_k16", kind="c", seed=0x5EED0101, n=3000, d=32, n_modes=10, k=16, iters=10, attempts=2),
    dict(name="km_n600_d300_k20", kind="c", seed=0x5EED0111, n=600, d=300, n_modes=6, k=20, iters=10, attempts=3),
    dict(name="km_n500_d8_k5_dup", kind="u", seed=0x5EED0121, n=500, d=8, dup=40, k=5, iters=6, attempts=1),
    dict(name="km_n1000_d768_k64", kind="c", seed=0x5EED0131, n=1000, d=768, n_modes=16, k=64, iters=4, attempts=1),
]


def init_draws(seed, attempts, k, n):
    """Stand-in for the reference's unseeded draws WITH replacement (ivfflat.rs:18-27)."""
    h = dg.mix64(np.uint64(seed) + np.arange(attempts * k, dtype=np.uint64))
    idx = (h % np.uint64(n)).astype(np.uint64)
    if k >= 3:
        idx[1] = idx[0]  # force a duplicate initial centroid (legal in the reference) -> empty cluster
    return idx


def make_kmeans():
    out = {}
    for cs in KMEANS_CASES:
        X = corpus(cs)
        k, n, d = cs["k"], cs["n"], cs["d"]
        init = init_draws(cs["seed"] ^ 0xABCD, cs["attempts"], k, n)
        # one assign + update + cost from the injected centroids (primitive kernels)
        C0 = X[init[:k].astype(np.int64)].copy()
        a_c, a_n = co.assign_to_clusters(X, C0), no.assign_to_clusters(X, C0)
        same(a_c, a_n, cs["name"] + " assign")
        u_c, u_n = co.update_centroids(X, a_c, k), no.update_centroids(X, a_n, k)
        same(u_c, u_n, cs["name"] + " update")
        c_c, c_n = co.kmeans_cost(X, C0, a_c), no.kmeans_cost(X, C0, a_n)
        same(np.array([c_c]), np.array([c_n]), cs["name"] + " cost")
        out[f"{cs['name']}/init"] = init
        out[f"{cs['name']}/assign0"] = a_c
        out[f"{cs['name']}/update0_bits"] = bits(u_c)
        out[f"{cs['name']}/cost0_bits"] = bits(np.array([c_c]))
        # full build_kmeans for the first draw
        Ck, ak, itk = co.build_kmeans(X, k, cs["iters"], init[:k])
        Cn, an, itn = no.build_kmeans(X, k, cs["iters"], init[:k])
        same(Ck, Cn, cs["name"] + " kmeans C"); same(ak, an, cs["name"] + " kmeans a")
        assert itk == itn
        out[f"{cs['name']}/kmeans_C_bits"] = bits(Ck)
        out[f"{cs['name']}/kmeans_assign"] = ak
        out[f"{cs['name']}/kmeans_iters"] = np.array([itk], dtype=np.uint64)
        # build_index, best of attempts
        bc = co.build_index(X, k, cs["attempts"], cs["iters"], init)
        bn = no.build_index(X, k, cs["attempts"], cs["iters"], init)
        same(bc["centroids"], bn["centroids"], cs["name"] + " build C")
        same(bc["assignments"], bn["assignments"], cs["name"] + " build a")
        same(np.array([bc["cost"]]), np.array([bn["cost"]]), cs["name"] + " build cost")
        out[f"{cs['name']}/build_C_bits"] = bits(bc["centroids"])
        out[f"{cs['name']}/build_assign"] = bc["assignments"]
        out[f"{cs['name']}/build_cost_bits"] = bits(np.array([bc["cost"]]))
        out[f"{cs['name']}/build_best_attempt"] = np.array([bc["best_attempt"]], dtype=np.uint64)
        out[f"{cs['name']}/crc"] = np.array([crc(X)], dtype=np.uint32)

        # search on the built index (+3 added vectors: ivfflat.rs:200-213)
        values = X.copy(); ids = [list(l) for l in bc["ids"]]; assign = list(bc["assignments"])
        extra = dg.dist_u(cs["seed"] + 7, 3, d)
        add_clusters = []
        for x in extra:
            cc, cn = co.add_cluster(bc["centroids"], x), no.add_cluster(bc["centroids"], x)
            assert cc == cn
            add_clusters.append(cc)
            ids[cc].append(len(assign)); assign.append(cc)
            values = np.concatenate([values, x[None]], axis=0)
        out[f"{cs['name']}/add_clusters"] = np.array(add_clusters, dtype=np.uint64)
        Q = queries(cs["seed"] + 3, 6, d, values)
        Q[1] = extra[1]  # an added row must retrieve itself at distance 0
        out[f"{cs['name']}/crc_q"] = np.array([crc(Q), crc(extra)], dtype=np.uint32)
        for top_k in (1, 10, 50):
            ri, rd, cnt = [], [], []
            for q in Q:
                ic, dc = co.search_approximate(values, bc["centroids"], ids, q, top_k)
                i_n, dn = no.search_approximate(values, bc["centroids"], ids, q, top_k)
                same(ic, i_n, cs["name"] + " search ids"); same(dc, dn, cs["name"] + " search dist")
                pad = top_k - len(ic)
                ri.append(np.concatenate([ic, np.full(pad, np.uint64(M64))]))
                rd.append(np.concatenate([bits(dc), np.zeros(pad, np.uint32)])); cnt.append(len(ic))
            out[f"{cs['name']}/search/k{top_k}/ids"] = np.stack(ri)
            out[f"{cs['name']}/search/k{top_k}/dist_bits"] = np.stack(rd)
            out[f"{cs['name']}/search/k{top_k}/count"] = np.array(cnt, dtype=np.uint32)
            for nprobe in (1, 4, k):
                ri, rd, cnt = [], [], []
                for q in Q:
                    ic, dc = co.search_nprobe(values, bc["centroids"], ids, q, top_k, nprobe)
                    i_n, dn = no.search_nprobe(values, bc["centroids"], ids, q, top_k, nprobe)
                    same(ic, i_n, cs["name"] + " nprobe ids"); same(dc, dn, cs["name"] + " nprobe dist")
                    pad = top_k - len(ic)
                    ri.append(np.concatenate([ic, np.full(pad, np.uint64(M64))]))
                    rd.append(np.concatenate([bits(dc), np.zeros(pad, np.uint32)])); cnt.append(len(ic))
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/ids"] = np.stack(ri)
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/dist_bits"] = np.stack(rd)
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/count"] = np.array(cnt, dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "kmeans_search.npz"), **out)
    print("kmeans_search.npz", len(out), "arrays")


COS_CASES = [
    dict(name="cos_n2500_d32_k16", kind="c", seed=0x5EED0201, n=2500, d=32, n_modes=10, k=16, iters=8, attempts=2),
    dict(name="cos_n700_d300_k20", kind="c", seed=0x5EED0211, n=700, d=300, n_modes=6, k=20, iters=6, attempts=1),
    dict(name="cos_n400_d8_k5_dup", kind="u", seed=0x5EED0221, n=400, d=8, dup=25, k=5, iters=5, attempts=1),
]


def make_cosdist():
    """The metric extension: IVFFlat with cosine distance 1 - dot (base.rs:153-155) wherever
    ivfflat.rs calls squared_euclidean -- build_index, add, search_approximate, and the nprobe extension."""
    out = {}
    for cs in COS_CASES:
        X = corpus(cs)
        k, n, d = cs["k"], cs["n"], cs["d"]
        init = init_draws(cs["seed"] ^ 0xABCD, cs["attempts"], k, n)
        bc = co.build_index(X, k, cs["attempts"], cs["iters"], init, metric=1)
        bn = no.build_index(X, k, cs["attempts"], cs["iters"], init, metric=1)
        same(bc["centroids"], bn["centroids"], cs["name"] + " build C")
        same(bc["assignments"], bn["assignments"], cs["name"] + " build a")
        same(np.array([bc["cost"]]), np.array([bn["cost"]]), cs["name"] + " build cost")
        out[f"{cs['name']}/init"] = init
        out[f"{cs['name']}/build_C_bits"] = bits(bc["centroids"])
        out[f"{cs['name']}/build_assign"] = bc["assignments"]
        out[f"{cs['name']}/build_cost_bits"] = bits(np.array([bc["cost"]]))
        out[f"{cs['name']}/crc"] = np.array([crc(X)], dtype=np.uint32)
        values = X.copy(); ids = [list(l) for l in bc["ids"]]; assign = list(bc["assignments"])
        extra = dg.dist_u(cs["seed"] + 7, 3, d)
        add_clusters = []
        for x in extra:
            cc, cn = co.add_cluster(bc["centroids"], x, metric=1), no.add_cluster(bc["centroids"], x, metric=1)
            assert cc == cn
            add_clusters.append(cc)
            ids[cc].append(len(assign)); assign.append(cc)
            values = np.concatenate([values, x[None]], axis=0)
        out[f"{cs['name']}/add_clusters"] = np.array(add_clusters, dtype=np.uint64)
        Q = queries(cs["seed"] + 3, 6, d, values)
        Q[1] = extra[1]
        out[f"{cs['name']}/crc_q"] = np.array([crc(Q), crc(extra)], dtype=np.uint32)
        for top_k in (1, 10, 40):
            for nprobe in (0, 1, 4, k):
                ri, rd, cnt = [], [], []
                for q in Q:
                    if nprobe == 0:
                        ic, dc = co.search_approximate(values, bc["centroids"], ids, q, top_k, metric=1)
                        i_n, dn = no.search_approximate(values, bc["centroids"], ids, q, top_k, metric=1)
                    else:
                        ic, dc = co.search_nprobe(values, bc["centroids"], ids, q, top_k, nprobe, metric=1)
                        i_n, dn = no.search_nprobe(values, bc["centroids"], ids, q, top_k, nprobe, metric=1)
                    same(ic, i_n, cs["name"] + " ids"); same(dc, dn, cs["name"] + " dist")
                    pad = top_k - len(ic)
                    ri.append(np.concatenate([ic, np.full(pad, np.uint64(M64))]))
                    rd.append(np.concatenate([bits(dc), np.zeros(pad, np.uint32)])); cnt.append(len(ic))
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/ids"] = np.stack(ri)
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/dist_bits"] = np.stack(rd)
                out[f"{cs['name']}/nprobe{nprobe}/k{top_k}/count"] = np.array(cnt, dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "ivf_cosdist.npz"), **out)
    print("ivf_cosdist.npz", len(out), "arrays")


M64 = 0xFFFFFFFFFFFFFFFF

if __name__ == "__main__":
    import sys as _sys
    if "--cosdist-only" not in _sys.argv:
        make_flat()
        make_kmeans()
    make_cosdist()
