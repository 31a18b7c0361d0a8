"""No caps where the reference has none (ivfflat.rs:166-195): top_k in {65, 100, 256} and nprobe = 128 against the oracle
bit for bit (results wider than one key per lane come 64 ranks per pass), the reference-mode spill walked through more
than 64 empty lists, and the cross-GPU merge of wide partial results."""
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


def check(ix, Q, top_k, nprobe, qs):
    ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
    for qi in qs:
        oi, od = (co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k) if nprobe == 0 else
                  co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe))
        assert cnt[qi] == len(oi), (top_k, nprobe, qi, cnt[qi], len(oi))
        assert np.array_equal(ids[qi, :len(oi)], oi), (top_k, nprobe, qi)
        assert np.array_equal(bits(dist[qi, :len(oi)]), bits(od)), (top_k, nprobe, qi)


def test_wide_top_k_and_many_probes():
    n, d, k = 9000, 48, 160
    X = dg.dist_c(0x701, n, d, 300, dg.default_sigma(d))
    X[100:140] = X[7]                                   # 41 identical rows: ties across a pass boundary decided by list position
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x701, 1, k, n))
    Q = dg.dist_c(0x702, 48, d, 300, dg.default_sigma(d)); Q[5] = X[7]
    for top_k in (65, 100, 256):
        for nprobe in (0, 3, 32, 128, 200):               # 200 > k: every list
            check(ix, Q, top_k, nprobe, range(0, 48, 5))
    check(ix, Q, 10, 128, range(0, 48, 7))                # many probes, narrow result
    check(ix, Q[:1], 100, 128, [0])                       # a single query through the generic planner (P > 64)
    check(ix, Q[:1], 100, 0, [0])
    # top_k larger than the number of reachable vectors: nprobe mode returns what there is
    ids, dist, cnt = ix.search_batch(Q[:4], 300, 1)
    for qi in range(4):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], 300, 1)
        assert cnt[qi] == len(oi) and np.array_equal(ids[qi, :len(oi)], oi)
    ix.close()


def test_many_probes_run_on_the_matrix_cores():
    """nprobe in (64, 1024]: the list scan stays on the matrix cores (fp16 shadow, exact finish) -- the work items carry (query, list)
    pairs, only the RANKING of more than 48 lists is exact, 64 ranks per pass (rounds 1-5 sent these shapes to the ordered chains at
    half the bytes per second).  Against the oracle bit for bit, and again with every certificate failing (the exact re-scan)."""
    n, d, k = 30000, 64, 320
    X = dg.dist_c(0x721, n, d, 900, dg.default_sigma(d))
    X[200:230] = X[11]                                   # ties decided by list position
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x721, 1, k, n))
    Q = dg.dist_c(0x722, 96, d, 900, dg.default_sigma(d)); Q[9] = X[11]
    try:
        for mode in (1, 2):
            capi.set_option("prescan", mode)
            for nprobe, top_k in ((65, 10), (128, 10), (128, 48), (256, 10), (300, 20), (1000, 10)):   # 1000 > k: every list
                st0 = ix.prescan_stats()
                check(ix, Q, top_k, nprobe, range(0, 96, 7))
                st1 = ix.prescan_stats()
                assert st1["batches"] - st0["batches"] == 1, (mode, nprobe, top_k)
                assert (st1["fallback_queries"] - st0["fallback_queries"] == 96) == (mode == 2), (mode, nprobe, top_k, st0, st1)
        # ... and the RANKING of 49 .. 200 lists stays on the matrix cores too (coarse_select_wide_kernel: candidates four keys per lane wide,
        # exact re-score, certificate); with every coarse certificate failing (coarse = 2) the query is re-ranked exactly inside the kernel
        capi.set_option("prescan", 1)
        for cmode in (0, 2):
            capi.set_option("coarse", cmode)
            for nprobe in (49, 65, 128, 200, 201):
                c0 = ix.coarse_stats()
                check(ix, Q, 10, nprobe, range(0, 96, 7))
                c1 = ix.coarse_stats()
                assert c1["mfma_batches"] - c0["mfma_batches"] == (1 if nprobe <= 200 else 0), (cmode, nprobe, c0, c1)
                if nprobe <= 200:
                    assert (c1["fallback_queries"] - c0["fallback_queries"] == 96) == (cmode == 2), (cmode, nprobe, c0, c1)
    finally:
        capi.set_option("prescan", 1); capi.set_option("coarse", 0)
    ix.close()


def test_wide_results_run_on_the_matrix_cores():
    """top_k in (48, 200]: candidate lists four keys per lane wide (wide.hip.h) through the matrix-core list scan's compactions and the
    exact finish (finish_wide.hip.h) -- rounds 1-5 sent these to the ordered chains, 64 ranks per pass.  Against the oracle bit for bit; with
    every certificate failing the exact re-scan emits 64 ranks per pass inside fallback_kernel; top_k = 201 stays on the ordered chains."""
    n, d, k = 30000, 64, 120
    X = dg.dist_c(0x731, n, d, 500, dg.default_sigma(d))
    X[300:420] = X[13]                                   # 121 identical rows: ties across every boundary, decided by list position
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x731, 1, k, n))
    Q = dg.dist_c(0x732, 72, d, 500, dg.default_sigma(d)); Q[9] = X[13]; Q[10] = X[13] * np.float32(1.0001)
    try:
        for mode in (1, 2):
            capi.set_option("prescan", mode)
            if mode == 2:   # (a fresh handle: a shadow that fails more than 1/8 of >= 256 queries retires itself, and the wide lists live on it)
                ix.close()
                ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x731, 1, k, n))
            shapes = ((49, 8), (64, 8), (65, 8), (100, 8), (128, 16), (200, 8), (100, 100), (150, 1)) if mode == 1 else ((65, 8), (128, 16), (200, 8))
            for top_k, nprobe in shapes:
                st0 = ix.prescan_stats()
                check(ix, Q, top_k, nprobe, range(0, 72, 5))
                st1 = ix.prescan_stats()
                assert st1["batches"] - st0["batches"] == 1, (mode, top_k, nprobe)
                assert (st1["fallback_queries"] - st0["fallback_queries"] == 72) == (mode == 2), (mode, top_k, nprobe, st0, st1)
            st0 = ix.prescan_stats()
            check(ix, Q, 201, 8, range(0, 72, 9))
            assert ix.prescan_stats()["batches"] == st0["batches"]
        capi.set_option("prescan", 1)
        capi.set_option("wide_k", 0)                      # the ordered chains again: same results
        st0 = ix.prescan_stats()
        check(ix, Q, 100, 8, range(0, 72, 9))
        assert ix.prescan_stats()["batches"] == st0["batches"]
    finally:
        capi.set_option("prescan", 1); capi.set_option("wide_k", 1)
    # a long-row index: the 16-query hi-only blocks carry the wide lists too
    n2, d2, k2 = 3000, 1536, 10
    X2 = dg.dist_c(0x733, n2, d2, 40, dg.default_sigma(d2))
    ix2 = IVFFlatIndex.build_index(k2, 1, 2, X2, init_indices=mg.init_draws(0x733, 1, k2, n2))
    Q2 = dg.dist_c(0x734, 24, d2, 40, dg.default_sigma(d2))
    st0 = ix2.prescan_stats()
    check(ix2, Q2, 100, 4, range(0, 24, 3))
    assert ix2.prescan_stats()["batches"] - st0["batches"] == 1
    ix.close(); ix2.close()


def test_reference_spill_through_more_than_64_empty_lists():
    """k = 150 centroids of which 140 are duplicates of one row -> their lists are empty (ties go to the lowest index) and
    ALL rank ahead of the far lists for queries near that row: the walk of ivfflat.rs:166-195 crosses > 64 empty lists."""
    n, d, k = 600, 16, 150
    X = dg.dist_c(0x711, n, d, 8, dg.default_sigma(d))
    init = mg.init_draws(0x711, 1, k, n)
    init[:140] = init[0]
    ix = IVFFlatIndex.build_index(k, 1, 0, X, init_indices=init)     # max_iterations = 0: the drawn centroids stay
    lens = ix.list_lengths()
    assert int((lens == 0).sum()) >= 130
    Q = np.stack([X[int(init[0])] + np.float32(1e-3) * dg.dist_u(0x712 + i, 1, d)[0] for i in range(6)]).astype(np.float32)
    for top_k in (10, 80, 590):
        check(ix, Q, top_k, 0, range(6))
    with pytest.raises(capi.VersError) as e:               # more results than vectors: the reference panics (index out of bounds)
        ix.search_batch(Q[:1], n + 1, 0)
    assert e.value.status == capi.ERR_INSUFFICIENT
    # the DEVICE-pointer entry cannot come back for a deeper ranking: it ranks as deep as the list lengths can make the walk
    # need (here: through all 140 empty lists) instead of latching INVALID as round 2 did past 48 lists
    import torch
    Qd = torch.from_numpy(Q).cuda()
    for top_k in (10, 80, 590):
        for b in (6, 1):
            ids = torch.zeros(b, top_k, dtype=torch.int64, device="cuda"); dist = torch.zeros(b, top_k, device="cuda"); cnt = torch.zeros(b, dtype=torch.int32, device="cuda")
            ix.search_dev(Qd.data_ptr(), d, b, top_k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
            ix.poll()
            gi, gd, gc = ids.cpu().numpy().astype(np.uint64), dist.cpu().numpy(), cnt.cpu().numpy()
            for qi in range(b):
                oi, od = co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k)
                assert gc[qi] == len(oi) and np.array_equal(gi[qi, :len(oi)], oi) and np.array_equal(bits(gd[qi, :len(oi)]), bits(od)), (top_k, b, qi)
    ids = torch.zeros(1, n + 1, dtype=torch.int64, device="cuda"); dist = torch.zeros(1, n + 1, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ix.search_dev(Qd.data_ptr(), d, 1, n + 1, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
    with pytest.raises(capi.VersError) as e:
        ix.poll()
    assert e.value.status == capi.ERR_INSUFFICIENT
    ix.close()


def test_cross_gpu_merge_of_wide_partials():
    import torch
    n, d, k, world, b = 4000, 32, 40, 3, 9
    X = dg.dist_c(0x721, n, d, 60, dg.default_sigma(d))
    init = mg.init_draws(0x721, 1, k, n)
    whole = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=init)
    shards = []
    for r in range(world):
        ix = IVFFlatIndex(d); ix.set_shard(r, world)
        ix.values, ix.centroids, ix.assignments = whole.values, whole.centroids, whole.assignments
        ix._upload(); shards.append(ix)
    Q = dg.dist_c(0x722, b, d, 60, dg.default_sigma(d))
    Qd = torch.from_numpy(Q).cuda()
    for nprobe, top_k in [(0, 100), (12, 150), (40, 70)]:
        keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda"); ids = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
        for r, ix in enumerate(shards):
            ix.search_partial_dev(Qd.data_ptr(), d, b, top_k, nprobe, keys[r].data_ptr(), ids[r].data_ptr()); ix.poll()
        oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda"); od = torch.zeros(b, top_k, device="cuda"); oc = torch.zeros(b, dtype=torch.int32, device="cuda")
        IVFFlatIndex.merge_partials_dev(keys.data_ptr(), ids.data_ptr(), b * top_k, world, b, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        torch.cuda.synchronize()
        gi, gd, gc = oi.cpu().numpy().astype(np.uint64), od.cpu().numpy(), oc.cpu().numpy()
        for q in range(b):
            o_i, o_d = (co.search_approximate(whole.values, whole.centroids, whole.ids, Q[q], top_k) if nprobe == 0 else
                        co.search_nprobe(whole.values, whole.centroids, whole.ids, Q[q], top_k, nprobe))
            assert gc[q] == len(o_i) and np.array_equal(gi[q, :len(o_i)], o_i) and np.array_equal(bits(gd[q, :len(o_i)]), bits(o_d)), (nprobe, top_k, q)
    for ix in shards:
        ix.close()
    whole.close()


def test_exhaustive_over_the_index_wider_than_a_wave():
    """vers_ivf_search_exhaustive* = utils::search_exhaustive (utils.rs:68-82) over the index's own values: any top_k, 64 ranks per
    pass; the sharded partial + merge path likewise (ranks split over three handles)."""
    import torch
    n, d, k = 2500, 24, 12
    X = dg.dist_c(0x731, n, d, 36, dg.default_sigma(d))
    X[40] = X[9]; X[1900] = X[9]
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x731, 1, k, n))
    ix.add(X[9].copy())                                          # a third duplicate, in an `add` slot
    vals = ix.values
    Q = np.concatenate([dg.dist_c(0x732, 3, d, 36, dg.default_sigma(d)), X[9:10]])
    for metric in (0, 1):
        for top_k in (65, 256, 1000):
            ids, dist, cnt = ix.search_exhaustive(Q, top_k, metric)
            for q in range(Q.shape[0]):
                oi, od = co.search_exhaustive(vals, Q[q], top_k, metric)
                assert cnt[q] == len(oi) and np.array_equal(ids[q, :len(oi)], oi) and np.array_equal(bits(dist[q, :len(oi)]), bits(od)), (metric, top_k, q)
    # sharded: every rank ranks its own rows, one gather + vers_topk_merge_dev gives the whole corpus's answer
    world, b, top_k = 3, Q.shape[0], 200
    shards = []
    for r in range(world):
        s = IVFFlatIndex(d); s.set_shard(r, world)
        s.values, s.centroids, s.assignments = ix.values, ix.centroids, ix.assignments
        s._upload(); shards.append(s)
    Qd = torch.from_numpy(Q).cuda()
    keys = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda"); idb = torch.empty(world, b, top_k, dtype=torch.int64, device="cuda")
    for r, s in enumerate(shards):
        s.search_exhaustive_partial_dev(Qd.data_ptr(), d, b, top_k, 0, keys[r].data_ptr(), idb[r].data_ptr()); s.poll()
    oi_ = torch.zeros(b, top_k, dtype=torch.int64, device="cuda"); od_ = torch.zeros(b, top_k, device="cuda"); oc_ = torch.zeros(b, dtype=torch.int32, device="cuda")
    IVFFlatIndex.merge_partials_dev(keys.data_ptr(), idb.data_ptr(), b * top_k, world, b, top_k, 1, oi_.data_ptr(), od_.data_ptr(), oc_.data_ptr())
    torch.cuda.synchronize()
    gi, gd, gc = oi_.cpu().numpy().astype(np.uint64), od_.cpu().numpy(), oc_.cpu().numpy()
    for q in range(b):
        oi, od = co.search_exhaustive(vals, Q[q], top_k, 0)
        assert gc[q] == len(oi) and np.array_equal(gi[q, :len(oi)], oi) and np.array_equal(bits(gd[q, :len(oi)]), bits(od)), q
    for s in shards:
        s.close()
    ix.close()


@pytest.mark.parametrize("metric", [0, 1], ids=["l2sq", "cosdist"])
def test_single_query_coarse_over_several_phases_of_chunks(metric):
    """coarse1_kernel walks a centroid tile in phases of 16 chunks of 32 columns: d = 1100 is three phases (the register
    buffers alternate A/B/A) with a partial last one, k = 130 three tiles with a partial last one; d = 40 a single phase of
    two chunks; NaN in a query must still latch the reference's panic (partial_cmp().unwrap())."""
    for n, d, k in ((1800, 1100, 130), (700, 40, 70), (300, 513, 3)):
        X = dg.dist_c(0x9100 + d, n, d, max(2, k // 2), dg.default_sigma(d))
        if metric:
            X = (X * (0.5 + (np.arange(n) % 5)[:, None] * 0.375)).astype(np.float32)
        ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x9100 + d, 1, k, n), metric=metric)
        Q = dg.dist_c(0x9200 + d, 6, d, max(2, k // 2), dg.default_sigma(d)); Q[3] = X[11]
        for top_k in (1, 10, 64):
            for nprobe in (0, 1, min(k, 7), min(k, 64)):
                for qi in range(6):
                    oi, od = (co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k, metric=metric) if nprobe == 0 else
                              co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe, metric=metric))
                    i1, d1, c1 = ix.search_batch(Q[qi], top_k, nprobe)   # one query: the fused coarse quantiser + plan
                    assert c1[0] == len(oi), (d, top_k, nprobe, qi)
                    assert np.array_equal(i1[0, :len(oi)], oi), (d, top_k, nprobe, qi)
                    assert np.array_equal(bits(d1[0, :len(oi)]), bits(od)), (d, top_k, nprobe, qi)
        q = Q[0].copy(); q[d - 1] = np.nan                                # (the last column: the last chunk of the last phase)
        with pytest.raises(capi.VersError) as e:
            ix.search_batch(q, 3, 4)
        assert e.value.status == capi.ERR_NAN
        i1, d1, c1 = ix.search_batch(Q[3], 1, 1)                         # the handle is usable afterwards
        assert d1[0, 0] == (np.float32(0.0) if metric == 0 else d1[0, 0]) and c1[0] == 1
        ix.close()
