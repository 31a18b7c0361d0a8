"""The exchange steps of the multi-GPU paths, over torch.distributed (backend "nccl" = RCCL over xGMI on the
GPUs; gloo in the tests).

Search: ONE all-gather of per-rank partial top-k per batch (`all_gather_partials`).
  part[0] = keys, part[1] = vec ids, each [b, top_k] int64/uint64 bit patterns (kKeyMax padded).  The
  gathered [world, 2, b, top_k] buffer feeds vers_topk_merge_dev with rank_stride = 2*b*top_k.
  b=1024, top_k=10: 160 KiB per rank -- latency-bound on the 7 x 153 GB/s xGMI links, one collective
  per batch, not a per-list or per-query exchange.

Row-sharded build_index: `TorchComm` is the `vers_comm_t` of include/vers_hip.h -- five synchronous callbacks
over raw device buffers (all_gather, send, recv, broadcast, all_to_all_v) that the library calls for the counts,
the chain of running centroid sums / the cost fold, the centroids and the rows-to-owner exchange.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch
import torch.distributed as dist


def all_gather_partials(part: "torch.Tensor", group=None) -> "torch.Tensor":
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
    try:
        dist.all_gather_into_tensor(out, part.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        chunks = [out[r] for r in range(world)]
        dist.all_gather(chunks, part.contiguous(), group=group)
    return out


# ---- vers_comm_t (include/vers_hip.h) ---------------------------------------------------------------------------
_AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_SR = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)
_BC = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)
_A2A = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


class VersComm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather", _AG), ("send", _SR),
                ("recv", _SR), ("broadcast", _BC), ("all_to_all_v", _A2A)]


class _Mem:
    """A raw buffer as something torch can wrap without copying."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """vers_comm_t over a torch.distributed process group.

    device = int: the pointers the library hands over are HBM pointers of that GPU.  With the nccl backend (RCCL) the
    collectives run on them in place; with gloo (tests: several ranks sharing one GPU) they are staged through host
    memory.  device = None: the pointers are host pointers (CPU-only tests of this class)."""

    def __init__(self, device, group=None):
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.calls = {"all_gather": 0, "send": 0, "recv": 0, "broadcast": 0, "all_to_all_v": 0}
        self.bytes = dict(self.calls)
        self.seconds = {k: 0.0 for k in self.calls}   # host wall clock inside each callback (incl. its synchronisation)
        self._cbs = (_AG(self._all_gather), _SR(self._send), _SR(self._recv), _BC(self._broadcast), _A2A(self._all_to_all_v))
        self.struct = VersComm(None, self.rank, self.world, *self._cbs)

    def ptr(self):
        return C.byref(self.struct)

    # -- buffers ---------------------------------------------------------------------------------------------------
    def _wrap(self, ptr, nbytes):
        """u8 tensor over [ptr, ptr+nbytes): device memory or (device=None) host memory."""
        if self.device is None:
            return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        return torch.as_tensor(_Mem(ptr, nbytes), device=torch.device("cuda", self.device))

    def _global_rank(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _staged(self):
        return self.device is not None and self.backend != "nccl"

    def _sync(self):
        if self.device is not None:
            torch.cuda.synchronize(self.device)

    def _guard(self, name, nbytes, fn):
        import time
        t0 = time.perf_counter()
        try:
            self.calls[name] += 1
            self.bytes[name] += int(nbytes)
            fn()
            self._sync()
            self.seconds[name] += time.perf_counter() - t0
            return 0
        except Exception as e:  # never unwind through the C frame
            print(f"[vers comm] {name} failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    # -- the five callbacks ------------------------------------------------------------------------------------------
    def _all_gather(self, _ctx, send_ptr, recv_ptr, nbytes):
        def run():
            if nbytes == 0:
                return
            s = self._wrap(send_ptr, nbytes); r = self._wrap(recv_ptr, nbytes * self.world)
            if self._staged():
                hs = s.cpu(); hr = torch.empty(nbytes * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(hr, hs, group=self.group)
                r.copy_(hr)
            else:
                dist.all_gather_into_tensor(r, s, group=self.group)
        return self._guard("all_gather", nbytes, run)

    def _send(self, _ctx, ptr, nbytes, peer):
        def run():
            if nbytes == 0:
                return
            t = self._wrap(ptr, nbytes)
            dist.send(t.cpu() if self._staged() else t, self._global_rank(peer), group=self.group)
        return self._guard("send", nbytes, run)

    def _recv(self, _ctx, ptr, nbytes, peer):
        def run():
            if nbytes == 0:
                return
            t = self._wrap(ptr, nbytes)
            if self._staged():
                h = torch.empty(nbytes, dtype=torch.uint8)
                dist.recv(h, self._global_rank(peer), group=self.group)
                t.copy_(h)
            else:
                dist.recv(t, self._global_rank(peer), group=self.group)
        return self._guard("recv", nbytes, run)

    def _broadcast(self, _ctx, ptr, nbytes, root):
        def run():
            if nbytes == 0:
                return
            t = self._wrap(ptr, nbytes)
            if self._staged():
                h = t.cpu()
                dist.broadcast(h, self._global_rank(root), group=self.group)
                if self.rank != root:
                    t.copy_(h)
            else:
                dist.broadcast(t, self._global_rank(root), group=self.group)
        return self._guard("broadcast", nbytes, run)

    def _all_to_all_v(self, _ctx, send_ptr, send_bytes, send_off, recv_ptr, recv_bytes, recv_off):
        W = self.world
        sb = [int(send_bytes[i]) for i in range(W)]; so = [int(send_off[i]) for i in range(W)]
        rb = [int(recv_bytes[i]) for i in range(W)]; ro = [int(recv_off[i]) for i in range(W)]

        def run():
            s_tot = max((so[i] + sb[i] for i in range(W)), default=0); r_tot = max((ro[i] + rb[i] for i in range(W)), default=0)
            on_gpu = self.device is not None
            rccl = self.backend == "nccl" and on_gpu
            # a rank with nothing to send (or receive) still takes part in the collective: on RCCL its empty operand must be a
            # DEVICE tensor of the collective's dtype like everybody else's (a CPU u8 placeholder is rejected on that rank
            # only, and its peers then hang inside the collective)
            dev = torch.device("cuda", self.device) if on_gpu else torch.device("cpu")
            empty = torch.empty(0, dtype=torch.uint8, device=dev if rccl else torch.device("cpu"))
            s = self._wrap(send_ptr, s_tot) if s_tot else empty
            r = self._wrap(recv_ptr, r_tot) if r_tot else empty
            contiguous = all(so[i] == sum(sb[:i]) for i in range(W)) and all(ro[i] == sum(rb[:i]) for i in range(W))
            if rccl and contiguous and all(x % 4 == 0 for x in sb + rb):
                # RCCL all-to-all in 4-byte words (rows are f32, ids u32): one collective over xGMI
                dist.all_to_all_single(r.view(torch.float32), s.view(torch.float32), [x // 4 for x in rb], [x // 4 for x in sb],
                                       group=self.group)
                return
            hs = s.cpu() if self._staged() else s
            hr = torch.empty(r_tot, dtype=torch.uint8) if self._staged() else r
            ops = []
            for p in range(W):
                if p == self.rank:
                    if sb[p]:
                        hr[ro[p]:ro[p] + rb[p]].copy_(hs[so[p]:so[p] + sb[p]])
                    continue
                if rb[p]:
                    ops.append(dist.P2POp(dist.irecv, hr[ro[p]:ro[p] + rb[p]], self._global_rank(p), group=self.group))
                if sb[p]:
                    ops.append(dist.P2POp(dist.isend, hs[so[p]:so[p] + sb[p]].contiguous(), self._global_rank(p), group=self.group))
            # ONE batch: RCCL fuses the sends and receives of a batch into a single grouped launch; issued one by one, both
            # sides' receive kernels can queue ahead of their sends and wait for each other.  (gloo: plain non-blocking pairs.)
            if ops:
                for q in dist.batch_isend_irecv(ops):
                    q.wait()
            if self._staged() and r_tot:
                r.copy_(hr)
        return self._guard("all_to_all_v", sum(sb), run)


# ---- vers_gather_t (include/vers_hip.h): the ONE exchange of vers_ivf_search_sharded_dev --------------------------------------
_GA = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class VersGather(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather_async", _GA)]


class TorchGather:
    """vers_gather_t over a torch.distributed process group -- the stand-in for libvers_rccl.so's ncclAllGather where RCCL
    cannot run (tests: several ranks sharing one GPU over gloo; CPU-only runs with device=None).  NOT stream-ordered like the
    real one: it waits for `stream`, moves the bytes through host memory and returns with them in place -- same results,
    none of the overlap.  backend nccl: all_gather_into_tensor on the raw device pointers (torch's own stream + a wait)."""

    def __init__(self, device, group=None):
        self.device, self.group = device, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.calls = 0
        self._cb = _GA(self._gather)
        self.struct = VersGather(None, self.rank, self.world, self._cb)

    def ptr(self):
        return C.cast(C.byref(self.struct), C.c_void_p)

    def _gather(self, _ctx, send_ptr, recv_ptr, nbytes, stream):
        try:
            self.calls += 1
            if self.device is None:  # host pointers (CPU tests of the protocol)
                s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send_ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)))
                r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv_ptr, C.POINTER(C.c_uint8)), shape=(nbytes * self.world,)))
                dist.all_gather_into_tensor(r, s, group=self.group)
                return 0
            dev = torch.device("cuda", self.device)
            ext = torch.cuda.ExternalStream(int(stream or 0), device=dev) if stream else torch.cuda.default_stream(dev)
            s = torch.as_tensor(_Mem(send_ptr, nbytes), device=dev); r = torch.as_tensor(_Mem(recv_ptr, nbytes * self.world), device=dev)
            ext.synchronize()  # the partial search queued on `stream` has written the send buffer
            if self.backend == "nccl":
                dist.all_gather_into_tensor(r, s, group=self.group)
                torch.cuda.synchronize(dev)
            else:
                hr = torch.empty(nbytes * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(hr, s.cpu(), group=self.group)
                with torch.cuda.stream(ext):
                    r.copy_(hr)
                ext.synchronize()
            return 0
        except Exception as e:  # never unwind through the C frame
            print(f"[vers gather] failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 1
