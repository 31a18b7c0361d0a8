"""ctypes binding of oracle/libvers_oracle.so (the C restatement, vers_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from vers_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvers_oracle.so")

VO_ERR_NAN, VO_ERR_INSUFFICIENT, VO_ERR_EMPTY = -2, -3, -4


class OraclePanic(Exception):
    def __init__(self, code):
        super().__init__({-2: "NaN compared", -3: "insufficient vectors", -4: "empty centroids"}.get(code, str(code)))
        self.code = code


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vers_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        fp, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64)
        L = _lib
        L.vo_squared_euclidean.restype = C.c_float
        L.vo_squared_euclidean.argtypes = [fp, fp, C.c_size_t]
        L.vo_dot.restype = C.c_float
        L.vo_dot.argtypes = [fp, fp, C.c_size_t]
        L.vo_cosine_distance.restype = C.c_float
        L.vo_cosine_distance.argtypes = [fp, fp, C.c_size_t]
        L.vo_normalize.restype = None
        L.vo_normalize.argtypes = [fp, fp, C.c_size_t]
        L.vo_search_exhaustive.restype = C.c_int64
        L.vo_search_exhaustive.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint64, fp, C.c_uint64, C.c_int, u64p, fp]
        L.vo_assign.restype = C.c_int
        L.vo_assign.argtypes = [fp, C.c_uint64, fp, C.c_uint64, C.c_uint64, u64p]
        L.vo_update.restype = None
        L.vo_update.argtypes = [fp, C.c_uint64, u64p, C.c_uint64, C.c_uint64, fp]
        L.vo_cost.restype = C.c_float
        L.vo_cost.argtypes = [fp, C.c_uint64, fp, u64p, C.c_uint64]
        L.vo_kmeans.restype = C.c_int
        L.vo_kmeans.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p, fp, u64p, u64p]
        L.vo_build.restype = C.c_int
        L.vo_build.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p, fp, u64p,
                               fp, C.POINTER(C.c_int), u64p]
        L.vo_search.restype = C.c_int64
        L.vo_search.argtypes = [fp, fp, C.c_uint64, C.c_uint64, u64p, u64p, fp, C.c_uint64, u64p, fp]
        L.vo_search_nprobe.restype = C.c_int64
        L.vo_search_nprobe.argtypes = [fp, fp, C.c_uint64, C.c_uint64, u64p, u64p, fp, C.c_uint64, C.c_uint64, u64p, fp]
        L.vo_add_cluster.restype = C.c_int64
        L.vo_add_cluster.argtypes = [fp, C.c_uint64, C.c_uint64, fp]
        # the same with a metric argument (0 = squared L2 as the reference, 1 = cosine distance 1 - dot: extension)
        L.vo_assign_m.restype = C.c_int
        L.vo_assign_m.argtypes = L.vo_assign.argtypes + [C.c_int]
        L.vo_cost_m.restype = C.c_float
        L.vo_cost_m.argtypes = L.vo_cost.argtypes + [C.c_int]
        L.vo_kmeans_m.restype = C.c_int
        L.vo_kmeans_m.argtypes = L.vo_kmeans.argtypes + [C.c_int]
        L.vo_build_m.restype = C.c_int
        L.vo_build_m.argtypes = L.vo_build.argtypes + [C.c_int]
        L.vo_search_m.restype = C.c_int64
        L.vo_search_m.argtypes = L.vo_search.argtypes + [C.c_int]
        L.vo_search_nprobe_m.restype = C.c_int64
        L.vo_search_nprobe_m.argtypes = L.vo_search_nprobe.argtypes + [C.c_int]
        L.vo_add_cluster_m.restype = C.c_int64
        L.vo_add_cluster_m.argtypes = L.vo_add_cluster.argtypes + [C.c_int]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def _chk(rc):
    if rc < 0:
        raise OraclePanic(int(rc))
    return rc


def squared_euclidean(a, b) -> np.float32:
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().vo_squared_euclidean(pa, pb, a.size))


def dot(a, b) -> np.float32:
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().vo_dot(pa, pb, a.size))


def normalize(a) -> np.ndarray:
    a, _ = _f(np.atleast_2d(a))
    out = np.empty_like(a)
    for i in range(a.shape[0]):
        lib().vo_normalize(a[i].ctypes.data_as(C.POINTER(C.c_float)), out[i].ctypes.data_as(C.POINTER(C.c_float)), a.shape[1])
    return out


def search_exhaustive(data, query, top_k, metric=0):
    data, pd = _f(data); query, pq = _f(query)
    n, d = data.shape
    ids = np.empty(max(1, min(top_k, n)), dtype=np.uint64); dist = np.empty(ids.size, dtype=np.float32)
    m = _chk(lib().vo_search_exhaustive(pd, n, d, d, pq, top_k, metric, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        dist.ctypes.data_as(C.POINTER(C.c_float))))
    return ids[:m], dist[:m]


def assign_to_clusters(X, Cn, metric=0):
    X, px = _f(X); Cn, pc = _f(Cn)
    out = np.empty(X.shape[0], dtype=np.uint64)
    _chk(lib().vo_assign_m(px, X.shape[0], pc, Cn.shape[0], X.shape[1], out.ctypes.data_as(C.POINTER(C.c_uint64)), metric))
    return out


def update_centroids(X, assign, k):
    X, px = _f(X); assign, pa = _u(assign)
    out = np.empty((k, X.shape[1]), dtype=np.float32)
    lib().vo_update(px, X.shape[0], pa, k, X.shape[1], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def kmeans_cost(X, Cn, assign, metric=0) -> np.float32:
    X, px = _f(X); Cn, pc = _f(Cn); assign, pa = _u(assign)
    return np.float32(lib().vo_cost_m(px, X.shape[0], pc, pa, X.shape[1], metric))


def build_kmeans(X, k, max_iterations, init_idx, metric=0):
    X, px = _f(X); init_idx, pi = _u(init_idx)
    n, d = X.shape
    Cn = np.empty((k, d), dtype=np.float32); a = np.empty(n, dtype=np.uint64); it = C.c_uint64(0)
    _chk(lib().vo_kmeans_m(px, n, d, k, max_iterations, pi, Cn.ctypes.data_as(C.POINTER(C.c_float)),
                           a.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(it), metric))
    return Cn, a, int(it.value)


def build_index(X, k, num_attempts, max_iterations, init_idx, metric=0):
    X, px = _f(X); init_idx, pi = _u(np.asarray(init_idx).reshape(-1))
    n, d = X.shape
    Cn = np.zeros((k, d), dtype=np.float32); a = np.zeros(n, dtype=np.uint64)
    cost = C.c_float(0); kept = C.c_int(0); best = C.c_uint64(0)
    _chk(lib().vo_build_m(px, n, d, k, num_attempts, max_iterations, pi, Cn.ctypes.data_as(C.POINTER(C.c_float)),
                          a.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(cost), C.byref(kept), C.byref(best), metric))
    if not kept.value:
        Cn = np.zeros((0, d), dtype=np.float32); a = np.zeros(0, dtype=np.uint64)
    ids = [np.nonzero(a == c)[0].astype(np.uint64) for c in range(k)]
    return dict(centroids=Cn, assignments=a, ids=ids, cost=np.float32(cost.value), kept=bool(kept.value),
                best_attempt=int(best.value))


def csr(ids):
    off = np.zeros(len(ids) + 1, dtype=np.uint64)
    for c, l in enumerate(ids):
        off[c + 1] = off[c] + len(l)
    flat = np.concatenate([np.asarray(l, dtype=np.uint64) for l in ids]) if len(ids) else np.zeros(0, np.uint64)
    return off, np.ascontiguousarray(flat, dtype=np.uint64)


def search_approximate(values, centroids, ids, query, top_k, metric=0):
    values, pv = _f(values); centroids, pc = _f(centroids); query, pq = _f(query)
    off, flat = csr(ids)
    oi = np.empty(max(1, top_k), dtype=np.uint64); od = np.empty(max(1, top_k), dtype=np.float32)
    m = _chk(lib().vo_search_m(pv, pc, centroids.shape[0], values.shape[1] if values.ndim == 2 else centroids.shape[1],
                               off.ctypes.data_as(C.POINTER(C.c_uint64)), flat.ctypes.data_as(C.POINTER(C.c_uint64)),
                               pq, top_k, oi.ctypes.data_as(C.POINTER(C.c_uint64)), od.ctypes.data_as(C.POINTER(C.c_float)), metric))
    return oi[:m], od[:m]


def search_nprobe(values, centroids, ids, query, top_k, nprobe, metric=0):
    values, pv = _f(values); centroids, pc = _f(centroids); query, pq = _f(query)
    off, flat = csr(ids)
    oi = np.empty(max(1, top_k), dtype=np.uint64); od = np.empty(max(1, top_k), dtype=np.float32)
    m = _chk(lib().vo_search_nprobe_m(pv, pc, centroids.shape[0], values.shape[1],
                                      off.ctypes.data_as(C.POINTER(C.c_uint64)), flat.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      pq, top_k, nprobe, oi.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      od.ctypes.data_as(C.POINTER(C.c_float)), metric))
    return oi[:m], od[:m]


def add_cluster(centroids, x, metric=0) -> int:
    centroids, pc = _f(centroids); x, px = _f(x)
    return int(_chk(lib().vo_add_cluster_m(pc, centroids.shape[0], centroids.shape[1], px, metric)))
