"""ctypes binding of libvers_rccl.so (include/vers_comm_rccl.h): the multi-GPU exchanges over an RCCL communicator the
library drives itself -- ONE ncclAllGather per search batch on the batch's own stream (vers_ivf_search_sharded_dev), the
five synchronous callbacks of the row-sharded build.  torch.distributed is used ONLY to hand the 128-byte communicator id
from rank 0 to the others (any backend; a Rust host would use its own channel)."""
from __future__ import annotations

import ctypes as C
import os

from . import capi
from .capi import _vp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VERS_RCCL_LIB_PATH") or os.path.join(_HERE, "lib", "libvers_rccl.so")
ID_BYTES = 128

_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class VersGather(C.Structure):
    """vers_gather_t (include/vers_hip.h)."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather_async", _GATHER_FN)]


# name -> (restype, argtypes); one entry per declaration in include/vers_comm_rccl.h
SIGNATURES = {
    "vers_rccl_last_error": (C.c_char_p, []),
    "vers_rccl_unique_id": (C.c_int32, [_vp]),
    "vers_rccl_create": (C.c_int32, [_vp, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(_vp)]),
    "vers_rccl_adopt": (C.c_int32, [_vp, C.c_int32, C.POINTER(_vp)]),
    "vers_rccl_destroy": (C.c_int32, [_vp]),
    "vers_rccl_abort": (C.c_int32, [_vp]),
    "vers_rccl_versions": (C.c_int32, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_uint64]),
    "vers_rccl_gather": (C.c_int32, [_vp, C.POINTER(VersGather)]),
    "vers_rccl_comm": (C.c_int32, [_vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m vers_amd.build`")
        capi.lib()  # one HIP runtime per process (capi preloads torch's when torch is installed); torch's librccl likewise
        try:
            import torch  # noqa: F401  (maps torch's librccl.so.1 first: one RCCL per process)
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status: int):
    if status != capi.OK:
        raise capi.VersError(status, lib().vers_rccl_last_error().decode("utf-8", "replace"))


def versions() -> dict:
    """RCCL the adapter was built against / runs on, and the shared object that provides it."""
    b, r = C.c_int32(0), C.c_int32(0)
    path = C.create_string_buffer(1024)
    check(lib().vers_rccl_versions(C.byref(b), C.byref(r), path, 1024))
    return {"built_against": b.value, "running_on": r.value, "librccl": path.value.decode("utf-8", "replace")}


class RcclComm:
    """An RCCL communicator owned by libvers_rccl.so.  `from_torch`: rank 0 makes the id, torch.distributed (whatever its
    backend) carries the 128 bytes to the other ranks, every rank joins.  `adopt`: around an ncclComm_t the host already has."""

    def __init__(self, id_bytes: bytes = None, rank: int = 0, world: int = 1, device: int = 0, adopt_comm_ptr: int = 0):
        self._h = _vp()
        self.rank, self.world, self.device = rank, world, device
        if adopt_comm_ptr:
            check(lib().vers_rccl_adopt(_vp(adopt_comm_ptr), device, C.byref(self._h)))
        else:
            assert len(id_bytes) == ID_BYTES
            buf = (C.c_char * ID_BYTES).from_buffer_copy(id_bytes)
            check(lib().vers_rccl_create(C.cast(buf, _vp), rank, world, device, C.byref(self._h)))
        self._gather = VersGather()
        check(lib().vers_rccl_gather(self._h, C.byref(self._gather)))
        from .dist import VersComm
        self._comm = VersComm()
        check(lib().vers_rccl_comm(self._h, C.cast(C.byref(self._comm), _vp)))
        self.rank, self.world = int(self._gather.rank), int(self._gather.world)

    @classmethod
    def adopt(cls, comm_ptr: int, device: int) -> "RcclComm":
        """Around an ncclComm_t the host already owns (vers_rccl_adopt), e.g. torch's: ProcessGroupNCCL._comm_ptr()."""
        return cls(adopt_comm_ptr=comm_ptr, device=device)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * ID_BYTES)()
        check(lib().vers_rccl_unique_id(C.cast(buf, _vp)))
        return bytes(buf)

    @classmethod
    def from_torch(cls, device: int, group=None) -> "RcclComm":
        """Collective-safe: every rank first loads the adapter (and rank 0 makes the id) and the ranks AGREE that all of them
        could (all_reduce MIN of an ok flag) before anyone enters the id's broadcast -- a rank whose library is missing or
        stale raises on every rank instead of leaving its peers inside broadcast_object_list; a second agreement follows
        ncclCommInitRank."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        flag_dev = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")

        def agree(ok: bool, what: str):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            if int(t.item()) != 1:
                raise capi.VersError(capi.ERR_COMM, f"{what} failed on at least one rank" + ("" if ok else " (this one)"))

        err, uid = None, None
        try:
            lib()
            if rank == 0:
                uid = cls.unique_id()
        except Exception as e:  # noqa: BLE001 -- whatever it is, the peers must hear about it before they block
            err = e
        try:
            agree(err is None, "loading libvers_rccl.so / ncclGetUniqueId")
        except capi.VersError:
            raise err if err is not None else capi.VersError(capi.ERR_COMM, "libvers_rccl.so could not be loaded on another rank")
        box = [uid]
        dist.broadcast_object_list(box, src=0, group=group)
        comm, err = None, None
        try:
            comm = cls(box[0], rank, world, device)
        except Exception as e:  # noqa: BLE001
            err = e
        try:
            agree(err is None, "ncclCommInitRank")
        except capi.VersError:
            if comm is not None:
                comm.abort()
            raise err if err is not None else capi.VersError(capi.ERR_COMM, "ncclCommInitRank failed on another rank")
        return comm

    def abort(self):
        """ncclCommAbort (vers_rccl_abort): after ANY rank returned non-zero from a sharded build / search."""
        if self._h:
            check(lib().vers_rccl_abort(self._h))

    def gather_ptr(self):
        """const vers_gather_t* for vers_ivf_search_sharded_dev."""
        return C.cast(C.byref(self._gather), _vp)

    def ptr(self):
        """const vers_comm_t* for vers_ivf_build_sharded_dev (same role as dist.TorchComm.ptr)."""
        return C.byref(self._comm)

    def close(self):
        if self._h:
            lib().vers_rccl_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
