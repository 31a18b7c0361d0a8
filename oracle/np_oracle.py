"""NumPy-float32 restatement of the vers IVFFlat hot path (second, independent oracle).

TEST INFRASTRUCTURE ONLY -- nothing under vers_amd/ may import this module.

It exists to pin oracle/vers_oracle.c: the reference (ashrielbrian/vers) has no
tests or golden vectors for this path ("parity unpinned")
and cannot be compiled here, so two restatements written independently from the
Rust source must agree BIT FOR BIT on every fixture before the fixture is
committed (tests/golden/make_golden.py), and again in tests/test_oracle_golden.py.

Sequential f32 accumulation is reproduced with ``np.add.accumulate(dtype=f32)``
(a strict left-to-right fold; ``np.sum`` is pairwise and must not be used).
All paths are relative to /root/reference/vers/src.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class RefPanic(Exception):
    """Stands for a Rust panic in the reference (unwrap on NaN / None, index OOB)."""


def _seqsum_rows(terms: np.ndarray) -> np.ndarray:
    """Row-wise strict left-to-right f32 sum starting from 0.0 (Iterator::sum)."""
    terms = np.ascontiguousarray(terms, dtype=f32)
    if terms.shape[-1] == 0:
        return np.zeros(terms.shape[:-1], dtype=f32)
    acc = np.add.accumulate(terms, axis=-1, dtype=f32)
    return acc[..., -1]


def squared_euclidean(a, b) -> np.ndarray:
    """indexes/base.rs:119-126 -- (a-b).powi(2) summed sequentially; broadcasts over rows."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    t = (a - b).astype(f32)
    return _seqsum_rows((t * t).astype(f32))


def dot(a, b) -> np.ndarray:
    """indexes/base.rs:91-93"""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    return _seqsum_rows((a * b).astype(f32))


def normalize(a) -> np.ndarray:
    """indexes/base.rs:95-105 (+ divide_by_scalar :74-83); rows independently."""
    a = np.atleast_2d(np.asarray(a, dtype=f32))
    m = np.sqrt(dot(a, a)).astype(f32)
    out = a.copy()
    big = ~(m < f32(1e-6))
    out[big] = (a[big] / m[big, None]).astype(f32)
    return out


def cosine_distance(a, b) -> np.ndarray:
    """indexes/base.rs:153-155 (normalized=true): 1 - dot."""
    return (f32(1.0) - dot(a, b)).astype(f32)


def _dist(a, b, metric=0) -> np.ndarray:
    """metric 0: squared_euclidean (what ivfflat.rs calls everywhere); 1: cosine distance 1 - dot (base.rs:153-155) in
    the same places -- the metric EXTENSION of SURVEY.md 8f-3, not a reference code path."""
    return squared_euclidean(a, b) if metric == 0 else cosine_distance(a, b)


def _stable_order(dist: np.ndarray) -> np.ndarray:
    """sorted_by(partial_cmp().unwrap()): stable ascending; NaN panics once compared."""
    if dist.size >= 2 and np.isnan(dist).any():
        raise RefPanic("partial_cmp().unwrap() on NaN")
    return np.argsort(dist, kind="stable")


def search_exhaustive(data, query, top_k, metric=0):
    """utils.rs:68-82"""
    data = np.asarray(data, dtype=f32)
    dist = squared_euclidean(data, query) if metric == 0 else cosine_distance(data, query)
    order = _stable_order(dist)[:top_k]
    return order.astype(np.uint64), dist[order]


def assign_to_clusters(X, C, metric=0) -> np.ndarray:
    """ivfflat.rs:29-46 -- first minimum wins (min_by); NaN panics when k >= 2."""
    X = np.asarray(X, dtype=f32)
    C = np.asarray(C, dtype=f32)
    if C.shape[0] == 0:
        if X.shape[0] == 0:
            return np.zeros(0, dtype=np.uint64)
        raise RefPanic("min_by on empty centroids -> unwrap on None")
    out = np.empty(X.shape[0], dtype=np.uint64)
    step = max(1, (1 << 22) // max(1, C.shape[0] * X.shape[1]))
    for s in range(0, X.shape[0], step):
        D = _dist(X[s:s + step, None, :], C[None, :, :], metric)  # [rows, k]
        if C.shape[0] >= 2 and np.isnan(D).any():
            raise RefPanic("NaN distance in assign_to_clusters")
        out[s:s + step] = np.argmin(D, axis=1)  # np.argmin returns the first minimum
    return out


def update_centroids(X, assign, k) -> np.ndarray:
    """ivfflat.rs:47-71 -- per cluster: 0.0 + x_i1 + x_i2 ... in ascending data order, / count."""
    X = np.asarray(X, dtype=f32)
    d = X.shape[1]
    out = np.zeros((k, d), dtype=f32)
    for c in range(k):
        members = np.nonzero(assign == c)[0]  # ascending
        if members.size == 0:
            continue  # :63-67 zero vector
        rows = np.concatenate([np.zeros((1, d), dtype=f32), X[members]], axis=0)
        s = np.add.accumulate(rows, axis=0, dtype=f32)[-1]
        out[c] = (s / f32(members.size)).astype(f32)
    return out


def kmeans_cost(X, C, assign, metric=0) -> np.float32:
    """ivfflat.rs:138-149 -- fold(0.0, +) over per-point distances in data order."""
    per_point = _dist(np.asarray(X, dtype=f32), np.asarray(C, dtype=f32)[assign], metric)
    if per_point.size == 0:
        return f32(0.0)
    return np.add.accumulate(per_point, dtype=f32)[-1]


def build_kmeans(X, k, max_iterations, init_idx, metric=0):
    """ivfflat.rs:73-100 with the unseeded draw (ivfflat.rs:18-27) injected as init_idx."""
    X = np.asarray(X, dtype=f32)
    C = X[np.asarray(init_idx, dtype=np.int64)].copy()
    iters = 0
    for _ in range(max_iterations):
        a = assign_to_clusters(X, C, metric)
        Cn = update_centroids(X, a, k)
        iters += 1
        if C.view(np.uint32).tobytes() == Cn.view(np.uint32).tobytes():  # to_hashkey equality
            break
        C = Cn
    return C, assign_to_clusters(X, C, metric), iters


def build_index(X, k, num_attempts, max_iterations, init_idx, metric=0):
    """ivfflat.rs:102-136.  Returns dict(centroids, assignments, ids, cost, kept)."""
    X = np.asarray(X, dtype=f32)
    best = f32(np.inf)
    bc = np.zeros((0, X.shape[1]), dtype=f32)
    ba = np.zeros(0, dtype=np.uint64)
    kept = False
    init_idx = np.asarray(init_idx).reshape(num_attempts, k) if num_attempts else init_idx
    for a in range(num_attempts):
        C, asg, _ = build_kmeans(X, k, max_iterations, init_idx[a], metric)
        cost = kmeans_cost(X, C, asg, metric)
        if cost < best:
            best, bc, ba, kept = cost, C, asg, True
    ids = [np.nonzero(ba == c)[0].astype(np.uint64) for c in range(k)]  # ascending vec_id
    return dict(centroids=bc, assignments=ba, ids=ids, cost=best, kept=kept)


def search_approximate(values, centroids, ids, query, top_k, metric=0):
    """ivfflat.rs:153-198 -- nearest list, spill to the next while short; concatenation."""
    values = np.asarray(values, dtype=f32)
    centroids = np.asarray(centroids, dtype=f32)
    cd = _dist(centroids, query, metric) if centroids.shape[0] else np.zeros(0, f32)
    ranked = _stable_order(cd)
    out_i, out_d = [], []
    curr, remainder = 0, top_k
    while len(out_i) < top_k:
        if curr >= len(ranked):
            raise RefPanic("index out of bounds: nearest_centroids[curr_cluster]")
        lst = np.asarray(ids[ranked[curr]], dtype=np.int64)
        dist = _dist(values[lst], query, metric) if lst.size else np.zeros(0, f32)
        order = _stable_order(dist)[:top_k]
        if order.size < remainder:
            out_i += list(lst[order]); out_d += list(dist[order])
            remainder -= order.size
            curr += 1
        else:
            out_i += list(lst[order[:remainder]]); out_d += list(dist[order[:remainder]])
            break
    return np.asarray(out_i, dtype=np.uint64), np.asarray(out_d, dtype=f32)


def search_nprobe(values, centroids, ids, query, top_k, nprobe, metric=0):
    """Extension (not in the reference): all rows of the nprobe nearest
    lists, one global stable sort over the concatenation in probe-rank order, take top_k."""
    values = np.asarray(values, dtype=f32)
    cd = _dist(np.asarray(centroids, dtype=f32), query, metric)
    ranked = _stable_order(cd)[:nprobe]
    cat = np.concatenate([np.asarray(ids[c], dtype=np.int64) for c in ranked]) if len(ranked) else np.zeros(0, np.int64)
    dist = _dist(values[cat], query, metric) if cat.size else np.zeros(0, f32)
    order = _stable_order(dist)[:top_k]
    return cat[order].astype(np.uint64), dist[order]


def add_cluster(centroids, x, metric=0) -> int:
    """ivfflat.rs:200-207 -- first-minimum centroid for an added vector."""
    centroids = np.asarray(centroids, dtype=f32)
    if centroids.shape[0] == 0:
        raise RefPanic("min_by on empty centroids -> unwrap on None")
    cd = _dist(centroids, x, metric)
    if cd.size >= 2 and np.isnan(cd).any():
        raise RefPanic("NaN distance in add")
    return int(np.argmin(cd))
