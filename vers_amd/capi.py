"""ctypes binding of libvers_hip.so (include/vers_hip.h).  No CPU fallback: if the HIP
library is missing this module raises at import of the symbol table."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VERS_LIB_PATH: another build of the same library -- same-box A/B runs of compile-time variants, scripts/ab_variant.sh)
LIB_PATH = os.environ.get("VERS_LIB_PATH") or os.path.join(_HERE, "lib", "libvers_hip.so")

OK, ERR_INVALID, ERR_NAN, ERR_INSUFFICIENT, ERR_HIP, ERR_EMPTY, ERR_COMM = 0, 1, 2, 3, 4, 5, 6
METRIC_L2SQ, METRIC_COSDIST = 0, 1
MAX_TOPK = 64


class VersError(RuntimeError):
    """A non-zero status from the C ABI.  The reference panics where status is NAN /
    INSUFFICIENT / EMPTY; host wrappers re-raise to keep that behaviour."""

    def __init__(self, status: int, msg: str):
        super().__init__(f"vers status {status}: {msg}")
        self.status = status


_lib = None
_fp, _u64p, _u32p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p

# name -> (restype, argtypes); one entry per declaration in include/vers_hip.h
SIGNATURES = {
    "vers_last_error": (C.c_char_p, []),
    "vers_abi_version": (C.c_int32, []),
    "vers_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "vers_flat_create": (C.c_int32, [C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "vers_flat_destroy": (C.c_int32, [_vp]),
    "vers_flat_upload": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "vers_flat_upload_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "vers_flat_search": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "vers_flat_search_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_flat_poll": (C.c_int32, [_vp, _vp]),
    "vers_flat_last_scan_ms": (C.c_int32, [_vp, C.POINTER(C.c_float)]),
    "vers_ivf_create": (C.c_int32, [C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "vers_ivf_destroy": (C.c_int32, [_vp]),
    "vers_ivf_set_metric": (C.c_int32, [_vp, C.c_uint32]),
    "vers_ivf_get_metric": (C.c_int32, [_vp, C.POINTER(C.c_uint32)]),
    "vers_ivf_build": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint64, _vp,
                                   C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp]),
    "vers_ivf_build_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint64, _vp,
                                       C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp]),
    "vers_ivf_upload": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp]),
    "vers_ivf_upload_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp]),
    "vers_ivf_upload_begin": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64]),
    "vers_ivf_upload_chunk": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64]),
    "vers_ivf_upload_chunk_dev": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64]),
    "vers_ivf_upload_end": (C.c_int32, [_vp]),
    "vers_ivf_add": (C.c_int32, [_vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vers_ivf_search": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "vers_ivf_search_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_ivf_poll": (C.c_int32, [_vp, _vp]),
    "vers_ivf_search_exhaustive": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "vers_ivf_search_exhaustive_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_ivf_info": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vers_ivf_list_lengths": (C.c_int32, [_vp, _vp]),
    "vers_ivf_last_scan": (C.c_int32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint32)]),
    "vers_shard_plan": (C.c_int32, [_vp, C.c_uint64, C.c_uint32, _vp]),
    "vers_ivf_set_shard": (C.c_int32, [_vp, C.c_uint32, C.c_uint32]),
    "vers_ivf_owners": (C.c_int32, [_vp, _vp]),
    "vers_ivf_coarse_ahead_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "vers_ivf_search_exhaustive_partial_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "vers_ivf_search_partial_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "vers_ivf_search_sharded_dev": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_ivf_search_exhaustive_sharded_dev": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_topk_merge_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "vers_ivf_last_coarse_ms": (C.c_int32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vers_ivf_last_finish_ms": (C.c_int32, [_vp, C.POINTER(C.c_float)]),
    "vers_ivf_coarse_stats": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vers_ivf_prescan_stats": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vers_ivf_shadow_state": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "vers_ivf_layout_bytes": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vers_ivf_build_sharded_dev": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                               C.c_uint64, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), _vp]),
    "vers_set_option": (C.c_int32, [C.c_char_p, C.c_int64]),
    "vers_mem_stats": (C.c_int32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]),
    "vers_ivf_scan_times": (C.c_int32, [_vp, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.c_int32]),
    "vers_ivf_get_list": (C.c_int32, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vers_ivf_get_centroids": (C.c_int32, [_vp, _vp, C.c_uint64]),
    "vers_gen_rows_dev": (C.c_int32, [_vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                      C.c_float, C.c_uint64, _vp]),
    "vers_kmeans_assign": (C.c_int32, [C.c_int32, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp]),
    "vers_kmeans_assign_dev": (C.c_int32, [C.c_int32, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp]),
    "vers_assign_stats": (C.c_int32, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]),
    "vers_build_stats": (C.c_int32, [C.POINTER(C.c_double), C.c_int32]),
    "vers_build_phases": (C.c_int32, [C.POINTER(C.c_double), C.c_int32]),
    "vers_kmeans_update": (C.c_int32, [C.c_int32, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint32, _vp]),
    "vers_kmeans_cost": (C.c_int32, [C.c_int32, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint32,
                                     C.POINTER(C.c_float)]),
}


def _preload_hip_runtime():
    """One HIP runtime per process: when PyTorch-ROCm is installed it bundles its own libamdhip64;
    loading /opt/rocm's copy first (through libvers_hip.so) and torch's later gives torch a second
    runtime that sees no GPU.  Map torch's copy first so both sides share it (torch is plumbing for
    device memory / streams / RCCL in tests and bench.py)."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m vers_amd.build` "
                              "(there is no CPU fallback for this path)")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def env_option(name: str, default: int) -> int:
    """What the library starts option `name` with in THIS environment: VERS_OPTIONS="name=value,..." (csrc/core.hip's table; the two
    documented switches VERS_SHADOW / VERS_ROWMAJOR are options "shadow" / "rowmajor"), else `default`.  vers_set_option overrides."""
    alias = {"shadow": "VERS_SHADOW", "rowmajor": "VERS_ROWMAJOR"}
    val = os.environ.get(alias[name]) if name in alias else None
    for kv in os.environ.get("VERS_OPTIONS", "").split(","):
        k, _, v = kv.partition("=")
        if k.strip() == name and v:
            val = v
    return default if val is None else int(val, 0)


def check(status: int):
    if status != OK:
        raise VersError(status, lib().vers_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int32(0)
    check(lib().vers_device_count(C.byref(n)))
    return n.value


def _ptr(a):
    return a.ctypes.data_as(_vp)


class FlatCorpus:
    """Brute-force scan over rows in HBM -- utils::search_exhaustive (utils.rs:68-82)."""

    def __init__(self, d: int, device: int = 0):
        self.d = int(d)
        self._h = _vp()
        check(lib().vers_flat_create(device, self.d, C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            lib().vers_flat_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, rows: np.ndarray):
        rows = np.asarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.d and rows.strides[1] == 4
        check(lib().vers_flat_upload(self._h, _ptr(rows), rows.shape[0], rows.strides[0] if rows.shape[0] else 4 * self.d))
        self.n = rows.shape[0]

    def upload_dev(self, data_ptr: int, n: int, ld: int):
        check(lib().vers_flat_upload_dev(self._h, _vp(data_ptr), n, ld))
        self.n = n

    def search(self, queries: np.ndarray, top_k: int, metric: int = METRIC_L2SQ):
        """-> (ids [b, top_k] u64, dist [b, top_k] f32, count [b] u32); rows beyond count are unspecified."""
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        b = q.shape[0]
        ids = np.zeros((b, max(top_k, 1)), dtype=np.uint64)
        dist = np.zeros((b, max(top_k, 1)), dtype=np.float32)
        cnt = np.zeros(b, dtype=np.uint32)
        check(lib().vers_flat_search(self._h, _ptr(q), 4 * q.shape[1], b, top_k, metric, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids[:, :top_k], dist[:, :top_k], cnt

    def search_dev(self, q_ptr: int, ldq: int, b: int, top_k: int, metric: int, ids_ptr: int, dist_ptr: int,
                   cnt_ptr: int, stream: int = 0):
        check(lib().vers_flat_search_dev(self._h, _vp(q_ptr), ldq, b, top_k, metric, _vp(ids_ptr), _vp(dist_ptr),
                                         _vp(cnt_ptr), _vp(stream)))

    def poll(self, stream: int = 0):
        check(lib().vers_flat_poll(self._h, _vp(stream)))

    def last_scan_ms(self) -> float:
        ms = C.c_float(0)
        check(lib().vers_flat_last_scan_ms(self._h, C.byref(ms)))
        return ms.value


# ---- k-means primitives (ivfflat.rs:29-71,138-149) -----------------------------------------
def _rows(a):
    a = np.asarray(a, dtype=np.float32)
    if a.ndim != 2 or (a.shape[0] and a.strides[1] != 4):
        a = np.ascontiguousarray(a, dtype=np.float32)
    return a, (a.strides[0] if a.shape[0] > 1 else 4 * a.shape[1])


def kmeans_assign(X, Cn, device: int = 0, want_min_dist: bool = False):
    X, sx = _rows(X); Cn, sc = _rows(Cn)
    out = np.zeros(X.shape[0], dtype=np.uint64)
    md = np.zeros(X.shape[0], dtype=np.float32) if want_min_dist else None
    check(lib().vers_kmeans_assign(device, _ptr(X), X.shape[0], sx, _ptr(Cn), Cn.shape[0], sc, X.shape[1], _ptr(out),
                                   _ptr(md) if want_min_dist else None))
    return (out, md) if want_min_dist else out


def kmeans_assign_dev(rows_ptr: int, n: int, ld: int, centroids_ptr: int, k: int, c_ld: int, d: int, out_assign_ptr: int,
                      out_min_dist_ptr: int = 0, device: int = 0):
    """assign_to_clusters (ivfflat.rs:29-46) on device-resident rows / centroids; out_assign u64 [n] on the device."""
    check(lib().vers_kmeans_assign_dev(device, _vp(rows_ptr), n, ld, _vp(centroids_ptr), k, c_ld, d, _vp(out_assign_ptr),
                                       _vp(out_min_dist_ptr) if out_min_dist_ptr else None))


def kmeans_assign_release(device: int = 0):
    """frees the device scratch vers_kmeans_assign_dev keeps between calls (a call with n == 0)"""
    check(lib().vers_kmeans_assign_dev(device, None, 0, 4, None, 0, 4, 4, None, None))


def kmeans_update(X, assign, k: int, device: int = 0):
    X, sx = _rows(X)
    assign = np.ascontiguousarray(assign, dtype=np.uint64)
    out = np.zeros((k, X.shape[1]), dtype=np.float32)
    check(lib().vers_kmeans_update(device, _ptr(X), X.shape[0], sx, _ptr(assign), k, X.shape[1], _ptr(out)))
    return out


def kmeans_cost(X, Cn, assign, device: int = 0) -> np.float32:
    X, sx = _rows(X); Cn, sc = _rows(Cn)
    assign = np.ascontiguousarray(assign, dtype=np.uint64)
    out = C.c_float(0)
    check(lib().vers_kmeans_cost(device, _ptr(X), X.shape[0], sx, _ptr(Cn), Cn.shape[0], sc, _ptr(assign), X.shape[1],
                                 C.byref(out)))
    return np.float32(out.value)


def gen_rows_dev(out_ptr: int, n: int, d: int, ld: int, kind: int, seed: int, seed_centres: int = 0, n_modes: int = 1,
                 sigma: float = 0.0, start_row: int = 0, stream: int = 0):
    check(lib().vers_gen_rows_dev(_vp(out_ptr), n, d, ld, kind, seed, seed_centres, n_modes, C.c_float(float(sigma)),
                                  start_row, _vp(stream)))


def shard_plan(list_lengths, world: int) -> np.ndarray:
    """Owner rank of every inverted list (deterministic LPT; host only, runs without a GPU)."""
    lens = np.ascontiguousarray(list_lengths, dtype=np.uint64)
    owner = np.zeros(max(lens.size, 1), dtype=np.uint8)
    check(lib().vers_shard_plan(_ptr(lens), lens.size, world, _ptr(owner)))
    return owner[:lens.size]


def assign_stats(reset=False):
    """(points assigned through the matrix-core path, of which re-done exactly) -- process-wide diagnostics."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().vers_assign_stats(C.byref(a), C.byref(b), 1 if reset else 0))
    return int(a.value), int(b.value)


def build_stats(reset=False) -> dict:
    """Where the build_index time of this process went (vers_build_stats; HIP events on the build's stream)."""
    v = (C.c_double * 8)()
    check(lib().vers_build_stats(v, 1 if reset else 0))
    keys = ("gemm_ms", "gemm_launches", "gemm_flop", "assign_ms", "assign_passes", "update_ms", "cost_ms", "redone_points")
    return dict(zip(keys, (float(x) for x in v)))


def build_phases(reset=False) -> dict:
    """build_index by phase, host wall clock in ms (vers_build_phases)."""
    v = (C.c_double * 10)()
    check(lib().vers_build_phases(v, 1 if reset else 0))
    keys = ("total_ms", "alloc_ms", "assign_ms", "assign_first_pass_ms", "assign_passes", "update_ms", "cost_ms", "install_lists_ms", "derive_ms", "other_ms")
    return dict(zip(keys, (round(float(x), 2) for x in v)))




def set_option(name: str, value: int):
    check(lib().vers_set_option(name.encode(), int(value)))


def mem_stats(reset_peak=False):
    """(bytes of device memory the library holds now, high-water mark since the last reset) -- process-wide."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(lib().vers_mem_stats(C.byref(a), C.byref(b), 1 if reset_peak else 0))
    return int(a.value), int(b.value)
