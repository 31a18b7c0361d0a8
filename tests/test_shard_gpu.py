"""Sharding by cluster: `world` handles on ONE GPU stand in for `world` processes.
Each keeps only its lists; partial top-k keys/ids are concatenated the way an all-gather would
and merged with vers_topk_merge_dev.  The result must equal the unsharded index / the oracle
bit for bit, in both the reference mode (nprobe=0, spill concatenation) and the nprobe mode."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_equals_unsharded(world):
    import torch
    n, d, k = 3000, 40, 24
    X = dg.dist_c(0x51, n, d, 30, dg.default_sigma(d))
    init = mg.init_draws(9, 1, k, n)
    whole = IVFFlatIndex.build_index(k, 1, 4, X, init_indices=init)
    shards = []
    for r in range(world):
        ix = IVFFlatIndex(d)
        ix.set_shard(r, world)
        # same deterministic build on every rank (as bench.py does), each keeps its own lists
        cent = np.zeros((k, d), np.float32); asg = np.zeros(n, np.uint64)
        import ctypes as C
        cost = C.c_float(0); kept = C.c_int32(0)
        capi.check(capi.lib().vers_ivf_build(ix._h, capi._ptr(X), n, 4 * d, k, 1, 4, capi._ptr(init), capi._ptr(cent), 4 * d,
                                             capi._ptr(asg), C.byref(cost), C.byref(kept), None))
        assert np.array_equal(asg, whole.assignments) and np.array_equal(bits(cent), bits(whole.centroids))
        shards.append(ix)
    owners = shards[0].owners()
    assert np.array_equal(owners, capi.shard_plan(whole.list_lengths(), world))
    for r in range(world):
        assert np.array_equal(shards[r].owners(), owners)
    # every rank adds the same two vectors: only the owner stores them, everyone counts them
    extra = dg.dist_u(77, 2, d)
    for x in extra:
        c0, v0 = whole.add(x)
        for ix in shards:
            c = capi.C.c_uint64(0); v = capi.C.c_uint64(0)
            capi.check(capi.lib().vers_ivf_add(ix._h, capi._ptr(np.ascontiguousarray(x)), capi.C.byref(c), capi.C.byref(v)))
            assert (c.value, v.value) == (c0, v0)
    b = 37
    Q = dg.dist_c(0x52, b, d, 30, dg.default_sigma(d)); Q[3] = extra[1]
    Qd = torch.from_numpy(Q).cuda()
    for nprobe, top_k in [(0, 10), (0, 64), (5, 10), (24, 33), (1, 1)]:
        keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
        ids = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
        for r, ix in enumerate(shards):
            ix.search_partial_dev(Qd.data_ptr(), d, b, top_k, nprobe, keys[r].data_ptr(), ids[r].data_ptr())
            ix.poll()
        oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda")
        od = torch.zeros(b, top_k, dtype=torch.float32, device="cuda")
        oc = torch.zeros(b, dtype=torch.int32, device="cuda")
        IVFFlatIndex.merge_partials_dev(keys.data_ptr(), ids.data_ptr(), b * top_k, world, b, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize()
        wi, wd, wc = whole.search_batch(Q, top_k, nprobe)
        gi, gd, gc = oi.cpu().numpy().astype(np.uint64), od.cpu().numpy(), oc.cpu().numpy()
        assert np.array_equal(gc, wc)
        for q in range(b):
            c = int(wc[q])
            assert np.array_equal(gi[q, :c], wi[q, :c]) and np.array_equal(bits(gd[q, :c]), bits(wd[q, :c])), (nprobe, top_k, q)
        # and against the oracle for a few queries
        for q in (0, 3, 36):
            o_i, o_d = (co.search_approximate(whole.values, whole.centroids, whole.ids, Q[q], top_k) if nprobe == 0 else
                        co.search_nprobe(whole.values, whole.centroids, whole.ids, Q[q], top_k, nprobe))
            assert np.array_equal(gi[q, :len(o_i)], o_i) and np.array_equal(bits(gd[q, :len(o_i)]), bits(o_d))
    # brute force (utils::search_exhaustive) over the row shards: same all-gather + merge, both metrics
    for metric, top_k in [(capi.METRIC_L2SQ, 10), (capi.METRIC_COSDIST, 7), (capi.METRIC_L2SQ, 64)]:
        keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
        ids = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
        for r, ix in enumerate(shards):
            ix.search_exhaustive_partial_dev(Qd.data_ptr(), d, b, top_k, metric, keys[r].data_ptr(), ids[r].data_ptr())
            ix.poll()
        oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda")
        od = torch.zeros(b, top_k, dtype=torch.float32, device="cuda")
        oc = torch.zeros(b, dtype=torch.int32, device="cuda")
        IVFFlatIndex.merge_partials_dev(keys.data_ptr(), ids.data_ptr(), b * top_k, world, b, top_k, 1, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize()
        wi, wd, wc = whole.search_exhaustive(Q, top_k, metric)
        assert np.array_equal(oc.cpu().numpy(), wc) and np.array_equal(oi.cpu().numpy().astype(np.uint64), wi)
        assert np.array_equal(bits(od.cpu().numpy()), bits(wd))
    for ix in shards:
        ix.close()
    whole.close()


def test_upload_dev_reshards_one_built_index_from_device_fields():
    """vers_ivf_upload_dev: the device cache from DEVICE-resident values / centroids / assignments (junk in the rows' padding
    columns), whole and with set_shard for worlds 2 and 4 -- every handle's lists, ids and search results equal the index
    build_index made (this is how scripts/emulate_shard.py walks every rank of every world over ONE build)."""
    import torch
    n, d, k, ld = 2500, 50, 20, 56
    X = dg.dist_c(0x61, n, d, 25, dg.default_sigma(d))
    init = mg.init_draws(5, 1, k, n)
    whole = IVFFlatIndex.build_index(k, 1, 5, X, init_indices=init)
    Xp = np.full((n, ld), np.nan, dtype=np.float32); Xp[:, :d] = X
    Xd = torch.from_numpy(Xp).cuda()
    Cd = torch.from_numpy(np.ascontiguousarray(whole.centroids)).cuda()
    Ad = torch.from_numpy(whole.assignments.astype(np.int64)).cuda()
    b, top_k = 21, 10
    Q = dg.dist_c(0x62, b, d, 25, dg.default_sigma(d))
    Qd = torch.from_numpy(Q).cuda()
    for world in (1, 2, 4):
        shards = []
        for r in range(world):
            ix = IVFFlatIndex(d)
            if world > 1:
                ix.set_shard(r, world)
            ix.upload_dev(Xd.data_ptr(), n, ld, Cd.data_ptr(), k, d, Ad.data_ptr())
            assert np.array_equal(ix.list_lengths(), whole.list_lengths())
            shards.append(ix)
        owners = shards[0].owners()
        for c in range(k):  # stored lists: rows and ids as the reference's ids[c] / values
            rows, ids = shards[int(owners[c])].get_list(c)
            assert np.array_equal(ids, np.asarray(whole.ids[c], dtype=np.uint64)) and np.array_equal(bits(rows), bits(X[ids.astype(np.int64)]))
        for nprobe in (0, 6):
            keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
            ids = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
            for r, ix in enumerate(shards):
                ix.search_partial_dev(Qd.data_ptr(), d, b, top_k, nprobe, keys[r].data_ptr(), ids[r].data_ptr())
                ix.poll()
            oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda")
            od = torch.zeros(b, top_k, dtype=torch.float32, device="cuda")
            oc = torch.zeros(b, dtype=torch.int32, device="cuda")
            IVFFlatIndex.merge_partials_dev(keys.data_ptr(), ids.data_ptr(), b * top_k, world, b, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
            torch.cuda.synchronize()
            wi, wd, wc = whole.search_batch(Q, top_k, nprobe)
            assert np.array_equal(oc.cpu().numpy(), wc)
            for q in range(b):
                c = int(wc[q])
                assert np.array_equal(oi.cpu().numpy().astype(np.uint64)[q, :c], wi[q, :c]) and np.array_equal(bits(od.cpu().numpy()[q, :c]), bits(wd[q, :c]))
        for ix in shards:
            ix.close()
    # an assignment out of range is refused
    Ad[7] = k
    ix = IVFFlatIndex(d)
    with pytest.raises(capi.VersError):
        ix.upload_dev(Xd.data_ptr(), n, ld, Cd.data_ptr(), k, d, Ad.data_ptr())
    ix.close(); whole.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_single_queries_and_tiny_batches(world):
    """b = 1 (the single query's own kernels: on the fp16 shadow since round 5, a rank scanning only ITS lists' records -- none
    at all when it owns nothing the query probes) and b = 2, 3 (consecutive single queries) through the partial search of every
    rank + the cross-rank merge, with both single-query scans: the unsharded index's results bit for bit."""
    import torch
    n, d, k = 4000, 72, 32
    X = dg.dist_c(0x71, n, d, 40, dg.default_sigma(d))
    whole = IVFFlatIndex.build_index(k, 1, 4, X, init_indices=mg.init_draws(6, 1, k, n))
    Xd = torch.from_numpy(X).cuda()
    Cd = torch.from_numpy(np.ascontiguousarray(whole.centroids)).cuda()
    Ad = torch.from_numpy(whole.assignments.astype(np.int64)).cuda()
    shards = []
    for r in range(world):
        ix = IVFFlatIndex(d); ix.set_shard(r, world)
        ix.upload_dev(Xd.data_ptr(), n, d, Cd.data_ptr(), k, d, Ad.data_ptr())
        shards.append(ix)
    Q = dg.dist_c(0x72, 9, d, 40, dg.default_sigma(d)); Q[4] = X[123]
    Qd = torch.from_numpy(Q).cuda()
    try:
        for single_shadow in (1, 0):
            capi.set_option("single_shadow", single_shadow)
            for b in (1, 2, 3):
                for nprobe, top_k in [(1, 10), (3, 1), (8, 10), (32, 30), (0, 10)]:
                    for q0 in (0, 3, 6):
                        keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
                        ids = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
                        for r, ix in enumerate(shards):
                            ix.search_partial_dev(Qd[q0:].data_ptr(), d, b, top_k, nprobe, keys[r].data_ptr(), ids[r].data_ptr())
                            ix.poll()
                        oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda"); od = torch.zeros(b, top_k, dtype=torch.float32, device="cuda")
                        oc = torch.zeros(b, dtype=torch.int32, device="cuda")
                        IVFFlatIndex.merge_partials_dev(keys.data_ptr(), ids.data_ptr(), b * top_k, world, b, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
                        torch.cuda.synchronize()
                        wi, wd, wc = whole.search_batch(Q[q0:q0 + b], top_k, nprobe)
                        assert np.array_equal(oc.cpu().numpy(), wc), (single_shadow, b, nprobe, top_k, q0)
                        for q in range(b):
                            c = int(wc[q])
                            assert np.array_equal(oi.cpu().numpy().astype(np.uint64)[q, :c], wi[q, :c]) and np.array_equal(bits(od.cpu().numpy()[q, :c]), bits(wd[q, :c])), (single_shadow, b, nprobe, top_k, q0, q)
    finally:
        capi.set_option("single_shadow", 1)
        for ix in shards:
            ix.close()
        whole.close()
