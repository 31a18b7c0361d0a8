"""Host-side mirror of vers's IVFFlatIndex (reference: vers/src/indexes/ivfflat.rs) on top of the
C ABI -- the Python stand-in for the Rust `impl Index<N> for IVFFlatIndexHip<N>` shown in
INTEGRATION.md.  Same method names, argument meaning and error behaviour as the reference: the
five fields stay host-owned (for save_index/load_index), the device handle is a cache.

A reference panic surfaces as capi.VersError (status NAN / INSUFFICIENT / EMPTY).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import capi
from .capi import _ptr, _vp, check, lib


class IVFFlatIndex:
    def __init__(self, d: int, device: int = 0, metric: int = capi.METRIC_L2SQ):
        """metric: METRIC_L2SQ = the reference's IVFFlat; METRIC_COSDIST = 1 - dot (base.rs:153-155) in every distance
        of build / add / search (extension, vers_ivf_set_metric)."""
        self.d = int(d)
        self.device = device
        self.metric = int(metric)
        self._h = _vp()
        check(lib().vers_ivf_create(device, self.d, C.byref(self._h)))
        if self.metric != capi.METRIC_L2SQ:
            check(lib().vers_ivf_set_metric(self._h, self.metric))
        # the reference's fields, in its order (ivfflat.rs:9-15)
        self.num_centroids = 0
        self.values = np.zeros((0, self.d), dtype=np.float32)
        self.centroids = np.zeros((0, self.d), dtype=np.float32)
        self.assignments = np.zeros(0, dtype=np.uint64)
        self.ids = []  # list of python lists of vec ids
        self.cost = np.float32(np.inf)
        self.iterations = None

    def close(self):
        if self._h:
            lib().vers_ivf_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- IVFFlatIndex::build_index (ivfflat.rs:102-136) -----------------------------------------
    @classmethod
    def build_index(cls, num_clusters: int, num_attempts: int, max_iterations: int, vectors, init_indices=None,
                    rng=None, device: int = 0, metric: int = capi.METRIC_L2SQ) -> "IVFFlatIndex":
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        n, d = vectors.shape
        self = cls(d, device, metric)
        if init_indices is None:
            # initialize_centroids (ivfflat.rs:18-27): k draws WITH replacement per attempt
            rng = rng or np.random.default_rng()
            init_indices = rng.integers(0, max(n, 1), size=num_attempts * num_clusters)
        init = np.ascontiguousarray(np.asarray(init_indices).reshape(-1), dtype=np.uint64)
        assert init.size == num_attempts * num_clusters
        cent = np.zeros((num_clusters, d), dtype=np.float32)
        asg = np.zeros(n, dtype=np.uint64)
        cost = C.c_float(0); kept = C.c_int32(0)
        iters = np.zeros(max(num_attempts, 1), dtype=np.uint64)
        check(lib().vers_ivf_build(self._h, _ptr(vectors), n, 4 * d, num_clusters, num_attempts, max_iterations,
                                   _ptr(init), _ptr(cent), 4 * d, _ptr(asg), C.byref(cost), C.byref(kept), _ptr(iters)))
        self.num_centroids = num_clusters
        self.values = vectors.copy()  # ivfflat.rs:131 vectors.clone()
        self.cost = np.float32(cost.value)
        self.iterations = iters[:num_attempts]
        if kept.value:
            self.centroids, self.assignments = cent, asg
        else:  # nothing beat +inf: empty centroids/assignments (ivfflat.rs:109-110)
            self.centroids = np.zeros((0, d), dtype=np.float32)
            self.assignments = np.zeros(0, dtype=np.uint64)
        self.ids = [[] for _ in range(num_clusters)]
        for vec_id, c in enumerate(self.assignments):  # ivfflat.rs:123-127
            self.ids[int(c)].append(vec_id)
        return self

    # -- Index::add (ivfflat.rs:200-213) ------------------------------------------------------------
    def add(self, embedding, vec_id: int = 0):
        """`vec_id` is accepted and ignored, exactly like the reference (ivfflat.rs:209)."""
        e = np.ascontiguousarray(embedding, dtype=np.float32).reshape(self.d)
        c = C.c_uint64(0); vid = C.c_uint64(0)
        check(lib().vers_ivf_add(self._h, _ptr(e), C.byref(c), C.byref(vid)))
        assert vid.value == len(self.assignments)
        self.values = np.concatenate([self.values, e[None]], axis=0)
        self.assignments = np.concatenate([self.assignments, np.array([c.value], dtype=np.uint64)])
        self.ids[c.value].append(int(vid.value))
        return int(c.value), int(vid.value)

    # -- Index::search_approximate (ivfflat.rs:153-198) ---------------------------------------------
    def search_approximate(self, query, top_k: int):
        ids, dist, cnt = self.search_batch(np.asarray(query, dtype=np.float32).reshape(1, self.d), top_k, nprobe=0)
        return [(int(i), np.float32(x)) for i, x in zip(ids[0, :cnt[0]], dist[0, :cnt[0]])]

    def search_batch(self, queries, top_k: int, nprobe: int = 0):
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        b = q.shape[0]
        ids = np.zeros((b, max(top_k, 1)), dtype=np.uint64)
        dist = np.zeros((b, max(top_k, 1)), dtype=np.float32)
        cnt = np.zeros(b, dtype=np.uint32)
        check(lib().vers_ivf_search(self._h, _ptr(q), 4 * self.d, b, top_k, nprobe, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids[:, :top_k], dist[:, :top_k], cnt

    def search_exhaustive(self, queries, top_k: int, metric: int = capi.METRIC_L2SQ):
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        b = q.shape[0]
        ids = np.zeros((b, max(top_k, 1)), dtype=np.uint64)
        dist = np.zeros((b, max(top_k, 1)), dtype=np.float32)
        cnt = np.zeros(b, dtype=np.uint32)
        check(lib().vers_ivf_search_exhaustive(self._h, _ptr(q), 4 * self.d, b, top_k, metric, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids[:, :top_k], dist[:, :top_k], cnt

    # -- device cache -----------------------------------------------------------------------------------
    def _upload(self):
        v = np.ascontiguousarray(self.values, dtype=np.float32)
        c = np.ascontiguousarray(self.centroids, dtype=np.float32)
        a = np.ascontiguousarray(self.assignments, dtype=np.uint64)
        check(lib().vers_ivf_upload(self._h, _ptr(v), v.shape[0], 4 * self.d, _ptr(c), c.shape[0], 4 * self.d, _ptr(a)))

    def info(self):
        n = C.c_uint64(0); k = C.c_uint64(0); m = C.c_uint64(0)
        check(lib().vers_ivf_info(self._h, C.byref(n), C.byref(k), C.byref(m)))
        return n.value, k.value, m.value

    def list_lengths(self):
        _, k, _ = self.info()
        out = np.zeros(max(k, 1), dtype=np.uint64)
        check(lib().vers_ivf_list_lengths(self._h, _ptr(out)))
        return out[:k]

    def last_scan(self):
        ms = C.c_float(0); u = C.c_uint64(0); s = C.c_uint64(0); it = C.c_uint32(0)
        check(lib().vers_ivf_last_scan(self._h, C.byref(ms), C.byref(u), C.byref(s), C.byref(it)))
        return dict(ms=ms.value, union_rows=u.value, streamed_rows=s.value, items=it.value)

    def last_coarse_ms(self):
        a = C.c_float(0); b = C.c_float(0)
        check(lib().vers_ivf_last_coarse_ms(self._h, C.byref(a), C.byref(b)))
        return dict(gemm_ms=a.value, select_ms=b.value)

    def coarse_stats(self):
        a = C.c_uint64(0); b = C.c_uint64(0)
        check(lib().vers_ivf_coarse_stats(self._h, C.byref(a), C.byref(b)))
        return dict(mfma_batches=a.value, fallback_queries=b.value)

    def prescan_stats(self):
        a = C.c_uint64(0); b = C.c_uint64(0)
        check(lib().vers_ivf_prescan_stats(self._h, C.byref(a), C.byref(b)))
        return dict(batches=a.value, fallback_queries=b.value)

    def last_finish_ms(self):
        ms = C.c_float(0)
        check(lib().vers_ivf_last_finish_ms(self._h, C.byref(ms)))
        return ms.value

    def layout_bytes(self):
        r = C.c_uint64(0); s_ = C.c_uint64(0); m = C.c_uint64(0)
        check(lib().vers_ivf_layout_bytes(self._h, C.byref(r), C.byref(s_), C.byref(m)))
        return dict(rows=r.value, shadow=s_.value, rowmajor=m.value)

    def shadow_state(self):
        a = C.c_int32(0); b = C.c_uint64(0)
        check(lib().vers_ivf_shadow_state(self._h, C.byref(a), C.byref(b)))
        return dict(active=bool(a.value), bytes=int(b.value))

    def scan_times(self, reset: bool = True):
        ms = np.zeros(1024, dtype=np.float32); n = C.c_uint32(0)   # (a ring of 64 per workspace; one workspace per stream in flight)
        check(lib().vers_ivf_scan_times(self._h, _ptr(ms), 1024, C.byref(n), 1 if reset else 0))
        return ms[:n.value].copy()

    def get_list(self, cluster: int):
        ln = C.c_uint64(0)
        check(lib().vers_ivf_get_list(self._h, cluster, None, 0, None, 0, C.byref(ln)))
        rows = np.zeros((ln.value, self.d), dtype=np.float32); ids = np.zeros(ln.value, dtype=np.uint64)
        check(lib().vers_ivf_get_list(self._h, cluster, _ptr(rows), 4 * self.d, _ptr(ids), ln.value, C.byref(ln)))
        return rows, ids

    def get_centroids(self):
        _, k, _ = self.info()
        c = np.zeros((k, self.d), dtype=np.float32)
        check(lib().vers_ivf_get_centroids(self._h, _ptr(c), 4 * self.d))
        return c

    # device-resident entry points (bench.py): torch tensors are passed as raw pointers
    def build_dev(self, rows_ptr: int, n: int, num_clusters: int, num_attempts: int, max_iterations: int, init_indices,
                  want_fields: bool = False):
        """build_index on rows already in HBM.  want_fields: also bring `centroids` and `assignments` (the reference's
        fields, ivfflat.rs:11-13) back to the host -- what upload_dev / save_index need."""
        init = np.ascontiguousarray(np.asarray(init_indices).reshape(-1), dtype=np.uint64)
        cost = C.c_float(0); kept = C.c_int32(0)
        iters = np.zeros(max(num_attempts, 1), dtype=np.uint64)
        ld = (self.d + 3) // 4 * 4
        cent = np.zeros((num_clusters, self.d), dtype=np.float32) if want_fields else None
        asg = np.zeros(max(n, 1), dtype=np.uint64) if want_fields else None
        check(lib().vers_ivf_build_dev(self._h, _vp(rows_ptr), n, ld, num_clusters, num_attempts, max_iterations, _ptr(init),
                                       _ptr(cent) if want_fields else None, 4 * self.d if want_fields else 0,
                                       _ptr(asg) if want_fields else None, C.byref(cost), C.byref(kept), _ptr(iters)))
        self.num_centroids = num_clusters
        self.cost = np.float32(cost.value); self.iterations = iters[:num_attempts]
        if want_fields and kept.value:
            self.centroids, self.assignments = cent, asg[:n]
        return bool(kept.value)

    def upload_dev(self, rows_ptr: int, n: int, ld: int, centroids_ptr: int, k: int, c_ld: int, assignments_ptr: int):
        """Device cache from DEVICE-resident fields (vers_ivf_upload_dev): rows [n][ld] f32, centroids [k][c_ld] f32,
        assignments [n] u64.  With set_shard only the lists LPT deals to this rank are stored."""
        check(lib().vers_ivf_upload_dev(self._h, _vp(rows_ptr), n, ld, _vp(centroids_ptr), k, c_ld, _vp(assignments_ptr)))
        self.num_centroids = k

    # Streamed rebuild of the device cache (vers_ivf_upload_begin / _chunk / _end): the reference's load_index -> search
    # sequence (base.rs:45-58, utils.rs:140-148) for an index no single GPU holds -- fields arrive in chunks, a sharded
    # handle keeps only its own lists, nothing of n_total rows is ever allocated.
    def upload_begin(self, centroids, list_lengths, n_total: int):
        c = np.ascontiguousarray(centroids, dtype=np.float32).reshape(-1, self.d)
        ll = np.ascontiguousarray(list_lengths, dtype=np.uint64)
        assert ll.shape[0] == c.shape[0]
        check(lib().vers_ivf_upload_begin(self._h, _ptr(c), c.shape[0], 4 * self.d, _ptr(ll), n_total))
        self.num_centroids = c.shape[0]

    def upload_chunk(self, rows, assignments, first_vec_id: int):
        v = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.d)
        a = np.ascontiguousarray(assignments, dtype=np.uint64)
        assert a.shape[0] == v.shape[0]
        check(lib().vers_ivf_upload_chunk(self._h, _ptr(v), 4 * self.d, _ptr(a), first_vec_id, v.shape[0]))

    def upload_chunk_dev(self, rows_ptr: int, ld: int, assignments_ptr: int, first_vec_id: int, n: int):
        check(lib().vers_ivf_upload_chunk_dev(self._h, _vp(rows_ptr), ld, _vp(assignments_ptr), first_vec_id, n))

    def upload_end(self):
        check(lib().vers_ivf_upload_end(self._h))

    def search_dev(self, q_ptr: int, ldq: int, b: int, top_k: int, nprobe: int, ids_ptr: int, dist_ptr: int, cnt_ptr: int,
                   stream: int = 0):
        check(lib().vers_ivf_search_dev(self._h, _vp(q_ptr), ldq, b, top_k, nprobe, _vp(ids_ptr), _vp(dist_ptr), _vp(cnt_ptr),
                                        _vp(stream)))

    def search_exhaustive_dev(self, q_ptr: int, ldq: int, b: int, top_k: int, metric: int, ids_ptr: int, dist_ptr: int,
                              cnt_ptr: int, stream: int = 0):
        check(lib().vers_ivf_search_exhaustive_dev(self._h, _vp(q_ptr), ldq, b, top_k, metric, _vp(ids_ptr), _vp(dist_ptr),
                                                   _vp(cnt_ptr), _vp(stream)))

    # -- sharding by cluster, one process per GPU -----------------------------------------------------------
    def set_shard(self, rank: int, world: int):
        check(lib().vers_ivf_set_shard(self._h, rank, world))

    def build_sharded_dev(self, rows_ptr: int, n_local: int, ld: int, row_begin: int, n_total: int, num_clusters: int,
                          num_attempts: int, max_iterations: int, init_indices, comm, want_assignments: bool = False):
        """build_index over a ROW-SHARDED corpus (vers_ivf_build_sharded_dev): this process holds rows
        [row_begin, row_begin + n_local) of n_total in HBM; `comm` is a vers_amd.dist.TorchComm (RCCL on the GPUs).
        Bit-identical to the single-process build; afterwards the handle holds the lists LPT deals to comm.rank."""
        init = np.ascontiguousarray(np.asarray(init_indices).reshape(-1), dtype=np.uint64)
        cost = C.c_float(0); kept = C.c_int32(0)
        iters = np.zeros(max(num_attempts, 1), dtype=np.uint64)
        asg = np.zeros(max(n_local, 1), dtype=np.uint64) if want_assignments else None
        check(lib().vers_ivf_build_sharded_dev(self._h, _vp(rows_ptr), n_local, ld, row_begin, n_total, num_clusters, num_attempts,
                                               max_iterations, _ptr(init), comm.ptr(), _ptr(asg) if want_assignments else None,
                                               C.byref(cost), C.byref(kept), _ptr(iters)))
        self.num_centroids = num_clusters
        self.cost = np.float32(cost.value); self.iterations = iters[:num_attempts]
        if want_assignments:
            self.local_assignments = asg[:n_local]
        return bool(kept.value)

    def search_sharded_dev(self, gather_ptr, q_ptr: int, ldq: int, b: int, top_k: int, nprobe: int, ids_ptr: int, dist_ptr: int,
                           cnt_ptr: int, stream: int = 0):
        """search_approximate over lists sharded by cluster, end to end on `stream` (vers_ivf_search_sharded_dev): partial
        search -> ONE all-gather (gather_ptr: vers_amd.rccl.RcclComm.gather_ptr() = ncclAllGather on the stream, or
        dist.TorchGather.ptr()) -> merge.  No host synchronisation."""
        check(lib().vers_ivf_search_sharded_dev(self._h, gather_ptr, _vp(q_ptr), ldq, b, top_k, nprobe, _vp(ids_ptr), _vp(dist_ptr),
                                                _vp(cnt_ptr), _vp(stream)))

    def search_exhaustive_sharded_dev(self, gather_ptr, q_ptr: int, ldq: int, b: int, top_k: int, metric: int, ids_ptr: int,
                                      dist_ptr: int, cnt_ptr: int, stream: int = 0):
        check(lib().vers_ivf_search_exhaustive_sharded_dev(self._h, gather_ptr, _vp(q_ptr), ldq, b, top_k, metric, _vp(ids_ptr),
                                                           _vp(dist_ptr), _vp(cnt_ptr), _vp(stream)))

    def owners(self):
        _, k, _ = self.info()
        o = np.zeros(max(k, 1), dtype=np.uint8)
        check(lib().vers_ivf_owners(self._h, _ptr(o)))
        return o[:k]

    def coarse_ahead_dev(self, q_ptr: int, ldq: int, b: int, nprobe: int, stream: int = 0):
        """Stage and rank the NEXT batch's queries on the handle's side stream while the current batch is scanned
        (vers_ivf_coarse_ahead_dev); the following search of the same query block picks the result up."""
        check(lib().vers_ivf_coarse_ahead_dev(self._h, _vp(q_ptr), ldq, b, nprobe, _vp(stream)))

    def search_exhaustive_partial_dev(self, q_ptr: int, ldq: int, b: int, top_k: int, metric: int, keys_ptr: int, ids_ptr: int,
                                      stream: int = 0):
        """Brute force over this rank's rows as (key, vec_id) pairs for merge_partials_dev (rows shard with the lists)."""
        check(lib().vers_ivf_search_exhaustive_partial_dev(self._h, _vp(q_ptr), ldq, b, top_k, metric, _vp(keys_ptr), _vp(ids_ptr),
                                                           _vp(stream)))

    def search_partial_dev(self, q_ptr: int, ldq: int, b: int, top_k: int, nprobe: int, keys_ptr: int, ids_ptr: int,
                           stream: int = 0):
        check(lib().vers_ivf_search_partial_dev(self._h, _vp(q_ptr), ldq, b, top_k, nprobe, _vp(keys_ptr), _vp(ids_ptr),
                                                _vp(stream)))

    @staticmethod
    def merge_partials_dev(keys_ptr: int, ids_ptr: int, rank_stride: int, world: int, b: int, top_k: int, nprobe: int,
                           out_ids_ptr: int, out_dist_ptr: int, out_cnt_ptr: int, stream: int = 0):
        check(lib().vers_topk_merge_dev(_vp(keys_ptr), _vp(ids_ptr), rank_stride, world, b, top_k, nprobe, _vp(out_ids_ptr),
                                        _vp(out_dist_ptr), _vp(out_cnt_ptr), _vp(stream)))

    def poll(self, stream: int = 0):
        check(lib().vers_ivf_poll(self._h, _vp(stream)))

    # -- Index::save_index / load_index (base.rs:31-58) ---------------------------------------------
    def save_index(self, file_path: str):
        write_index_file(file_path, self.num_centroids, self.values, self.centroids, self.assignments, self.ids)

    @classmethod
    def load_index(cls, file_path: str, d: int, device: int = 0, metric: int = capi.METRIC_L2SQ) -> "IVFFlatIndex":
        """`d` plays the role of the const generic N of IVFFlatIndex<N>.  (The file has no metric field -- the
        reference has no metric switch -- so a cosine-distance index is reloaded with metric=METRIC_COSDIST.)"""
        f = read_index_file(file_path, d)
        self = cls(d, device, metric)
        self.num_centroids, self.values, self.centroids = f["num_centroids"], f["values"], f["centroids"]
        self.assignments, self.ids = f["assignments"], f["ids"]
        self._upload()
        return self


# The index file of Index::save_index (base.rs:31-43): bincode 1.3.3 with default options over the five fields of
# IVFFlatIndex<N> in declaration order (ivfflat.rs:9-15) -- little endian, fixed-width integers, usize and every
# length as u64, a struct is its fields back to back without tags or names, Vec<T> = u64 length + items,
# Vector<N>([f32; N]) through serde_arrays = N raw f32 WITHOUT a length (a fixed-size array is a tuple to serde), and
# no padding: the 256-byte alignment of Vector<N> exists in memory only.  (bincode / serde_arrays sources are not
# vendored in the reference: layout restated from their published format -- "parity unpinned" for this row, see
# DESIGN.md; tests/test_index_file.py pins these rules byte by byte and against the C++ host mirror.)
def write_index_file(file_path, num_centroids, values, centroids, assignments, ids):
    with open(file_path, "wb") as f:
        f.write(struct.pack("<Q", int(num_centroids)))
        for mat in (values, centroids):
            m = np.ascontiguousarray(mat, dtype="<f4")
            f.write(struct.pack("<Q", m.shape[0])); f.write(m.tobytes())
        a = np.ascontiguousarray(assignments, dtype="<u8")
        f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())
        f.write(struct.pack("<Q", len(ids)))
        for lst in ids:
            l = np.asarray(lst, dtype="<u8")
            f.write(struct.pack("<Q", l.size)); f.write(l.tobytes())


def read_index_file(file_path, d: int) -> dict:
    """The five fields; raises IOError("Deserialization error: ...") like base.rs:52-57 on a short or oversized file."""
    with open(file_path, "rb") as f:
        buf = f.read()
    off = 0

    def u64():
        nonlocal off
        if off + 8 > len(buf):
            raise IOError("Deserialization error: unexpected end of file")
        (v,) = struct.unpack_from("<Q", buf, off); off += 8
        return v

    def take(dtype, count):
        nonlocal off
        nbytes = count * np.dtype(dtype).itemsize
        if off + nbytes > len(buf):
            raise IOError("Deserialization error: unexpected end of file")
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=off).copy(); off += nbytes
        return a

    out = {"num_centroids": u64()}
    out["values"] = take("<f4", u64() * d).reshape(-1, d)
    out["centroids"] = take("<f4", u64() * d).reshape(-1, d)
    out["assignments"] = take("<u8", u64())
    out["ids"] = [list(map(int, take("<u8", u64()))) for _ in range(u64())]
    return out
