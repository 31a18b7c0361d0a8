"""ctypes binding of libvers_hip_test.so (include/vers_hip_test.h): the TEST and one-GPU-emulation hooks.  They are NOT in the
product library (libvers_hip.so exports nothing named *test*): this second library links against it and takes its handles.
Used by tests/ and by the emulation / nominal-rank scripts only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi
from .capi import _ptr, _vp, check

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvers_hip_test.so")

# name -> (restype, argtypes); one entry per declaration in include/vers_hip_test.h
SIGNATURES = {
    "vers_ivf_test_poison_slack": (C.c_int32, [_vp, C.c_float]),
    "vers_ivf_test_last_vals": (C.c_int32, [_vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    "vers_test_mfma": (C.c_int32, [C.c_int32, C.c_uint32, _vp, _vp, C.c_uint32, _vp]),
    "vers_test_standin_gather": (C.c_int32, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vers_test_wave_net": (C.c_int32, [C.c_int32, _vp, _vp]),
    "vers_test_wide_net": (C.c_int32, [C.c_int32, _vp, _vp]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        capi.lib()  # the product library first: the hooks library resolves its internals against the copy already mapped
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m vers_amd.build`")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def poison_slack(index, value: float):
    """fill the storage rows of `index` (vers_amd.index.IVFFlatIndex) that hold no vector with `value` (what uninitialised memory may look like)"""
    check(lib().vers_ivf_test_poison_slack(index._h, C.c_float(value)))


def last_vals(index, q: int, cap: int = 8192):
    """(vec_ids, vals, per-candidate bounds, info dict) of query q of the last batched nprobe search on `index`"""
    ids = np.zeros(cap, dtype=np.uint64); vals = np.zeros(cap, dtype=np.float32); bnd = np.zeros(cap, dtype=np.float64)
    n = C.c_uint32(0); info = (C.c_double * 8)()
    check(lib().vers_ivf_test_last_vals(index._h, q, _ptr(ids), _ptr(vals), _ptr(bnd), cap, C.byref(n), info))
    m = min(n.value, cap)
    keys = ("qn", "xmax2", "r2", "bound_outside", "bound_common", "kp", "shadow", "metric")
    return ids[:m].copy(), vals[:m].copy(), bnd[:m].copy(), dict(zip(keys, (float(x) for x in info)))


def mfma(kind: int, A: np.ndarray, B: np.ndarray, device: int = 0) -> np.ndarray:
    """A [rows, K] x B [K, cols] on one wave of the matrix-core instruction `kind`, f32 result"""
    A = np.ascontiguousarray(A); B = np.ascontiguousarray(B)
    rows, cols = (64, 16) if kind == 3 else (32, 32)
    assert A.shape[0] == rows and B.shape[1] == cols and A.shape[1] == B.shape[0] and A.dtype == B.dtype
    assert A.dtype == (np.uint16 if kind <= 1 else np.float32)
    out = np.zeros((rows, cols), dtype=np.float32)
    check(lib().vers_test_mfma(device, kind, _ptr(A), _ptr(B), A.shape[1], _ptr(out)))
    return out


def wave_net(keys: np.ndarray, device: int = 0) -> np.ndarray:
    """the kernels' lane networks on 128 keys in one wave -> 640 words (include/vers_hip_test.h)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    assert keys.size == 128
    out = np.zeros(640, dtype=np.uint64)
    check(lib().vers_test_wave_net(device, _ptr(keys), _ptr(out)))
    return out


def wide_net(keys: np.ndarray, device: int = 0) -> np.ndarray:
    """the wide lists' networks (wide.hip.h) on 512 keys in one wave -> 771 words (include/vers_hip_test.h)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    assert keys.size == 512
    out = np.zeros(771, dtype=np.uint64)
    check(lib().vers_test_wide_net(device, _ptr(keys), _ptr(out)))
    return out


def standin_gather(gather_struct, rank: int, world: int, workgroups: int, spin_us: int, threads: int, lds_bytes: int):
    """fills a vers_gather_t (ctypes structure) with the exchange stand-in that has RCCL's footprint"""
    check(lib().vers_test_standin_gather(C.cast(C.byref(gather_struct), _vp), rank, world, workgroups, spin_us, threads, lds_bytes))
