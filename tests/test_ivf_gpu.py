"""GPU parity of build_index / add / search_approximate (vers_ivf_*) -- reads like the harness the
reference itself would run (utils.rs:117-184): build -> add -> save -> load -> search, checked
against the golden fixtures and the oracle.  Bar: bit-exact (ids, order, distance bits, centroid
bits, cost bits)."""
import os

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


def check_search(index, g, nm, Q, k_clusters, tag_fn):
    for top_k in (1, 10, 50):
        for tag, nprobe in [("search", 0), ("nprobe1", 1), ("nprobe4", 4), (f"nprobe{k_clusters}", k_clusters)]:
            gi = g[f"{nm}/{tag}/k{top_k}/ids"]; gd = g[f"{nm}/{tag}/k{top_k}/dist_bits"]; gc = g[f"{nm}/{tag}/k{top_k}/count"]
            ids, dist, cnt = index.search_batch(Q, top_k, nprobe)          # batched
            assert np.array_equal(cnt, gc), (tag, top_k)
            for qi in range(Q.shape[0]):
                c = int(gc[qi])
                assert np.array_equal(ids[qi, :c], gi[qi, :c]), (tag, top_k, qi)
                assert np.array_equal(bits(dist[qi, :c]), gd[qi, :c]), (tag, top_k, qi)
            i1, d1, c1 = index.search_batch(Q[2], top_k, nprobe)           # single query path
            c = int(gc[2])
            assert c1[0] == c and np.array_equal(i1[0, :c], gi[2, :c]) and np.array_equal(bits(d1[0, :c]), gd[2, :c])


@pytest.mark.parametrize("cs", mg.KMEANS_CASES, ids=lambda c: c["name"])
def test_build_add_save_load_search_golden(cs, golden_km, tmp_path):
    g, nm = golden_km, cs["name"]
    X = mg.corpus(cs); k, n, d = cs["k"], cs["n"], cs["d"]
    init = g[nm + "/init"]
    index = IVFFlatIndex.build_index(k, cs["attempts"], cs["iters"], X, init_indices=init)
    assert np.array_equal(bits(index.centroids), g[nm + "/build_C_bits"])
    assert np.array_equal(index.assignments, g[nm + "/build_assign"])
    assert bits(np.array([index.cost]))[0] == g[nm + "/build_cost_bits"][0]
    assert np.array_equal(np.sort(np.concatenate([np.asarray(l, dtype=np.int64) for l in index.ids])), np.arange(n))
    assert np.array_equal(index.list_lengths(), np.array([len(l) for l in index.ids], dtype=np.uint64))
    # add three vectors; the caller's vec_id is ignored (ivfflat.rs:209)
    extra = dg.dist_u(cs["seed"] + 7, 3, d)
    for x, want in zip(extra, g[nm + "/add_clusters"]):
        c, vid = index.add(x, vec_id=123456)
        assert c == want and vid == len(index.assignments) - 1
    Q = mg.queries(cs["seed"] + 3, 6, d, index.values); Q[1] = extra[1]
    check_search(index, g, nm, Q, k, None)
    # save -> load -> search again (utils.rs:140-148): the device cache is rebuilt from the host fields
    path = os.path.join(tmp_path, "ivfflat.index")
    index.save_index(path)
    re = IVFFlatIndex.load_index(path, d)
    assert re.num_centroids == k and np.array_equal(re.assignments, index.assignments) and re.ids == index.ids
    check_search(re, g, nm, Q, k, None)
    # Index::search_approximate proper: list of (vec_id, squared_l2); self-retrieval at exactly 0.0
    r = re.search_approximate(index.values[n + 1], 1)
    assert r[0][0] == n + 1 and r[0][1] == 0.0
    index.close(); re.close()


def test_batched_grouping_many_queries_per_list():
    # 200 queries over 12 lists: every list is shared by many queries (QG=8 blocks, several groups per list)
    n, d, k = 4000, 64, 12
    X = dg.dist_c(0x77, n, d, 12, dg.default_sigma(d))
    init = mg.init_draws(5, 1, k, n)
    index = IVFFlatIndex.build_index(k, 1, 5, X, init_indices=init)
    Q = dg.dist_c(0x78, 200, d, 12, dg.default_sigma(d))
    for nprobe, top_k in [(0, 10), (3, 10), (12, 64), (1, 1)]:
        ids, dist, cnt = index.search_batch(Q, top_k, nprobe)
        for qi in range(0, 200, 7):
            if nprobe == 0:
                oi, od = co.search_approximate(index.values, index.centroids, index.ids, Q[qi], top_k)
            else:
                oi, od = co.search_nprobe(index.values, index.centroids, index.ids, Q[qi], top_k, nprobe)
            assert cnt[qi] == len(oi)
            assert np.array_equal(ids[qi, :len(oi)], oi) and np.array_equal(bits(dist[qi, :len(oi)]), bits(od))
    # recall ground truth path: exhaustive over the index's values == utils::search_exhaustive
    ids, dist, cnt = index.search_exhaustive(Q[:9], 10)
    for qi in range(9):
        oi, od = co.search_exhaustive(index.values, Q[qi], 10)
        assert np.array_equal(ids[qi], oi) and np.array_equal(bits(dist[qi]), bits(od))
    index.close()


def test_reference_error_semantics():
    d = 6
    X = dg.dist_u(1, 40, d)
    # num_attempts == 0 keeps nothing: empty centroids; search panics (ivfflat.rs:169), add panics (:207)
    empty = IVFFlatIndex.build_index(4, 0, 5, X, init_indices=np.zeros(0, dtype=np.uint64))
    assert empty.centroids.shape[0] == 0 and empty.assignments.size == 0 and len(empty.ids) == 4
    with pytest.raises(capi.VersError) as e:
        empty.search_approximate(X[0], 3)
    assert e.value.status == capi.ERR_INSUFFICIENT
    with pytest.raises(capi.VersError) as e:
        empty.add(X[0], 0)
    assert e.value.status == capi.ERR_EMPTY
    assert empty.search_approximate(X[0], 0) == []      # top_k == 0 never touches the centroids
    index = IVFFlatIndex.build_index(4, 1, 5, X, init_indices=np.array([0, 1, 2, 3], dtype=np.uint64))
    # more results than vectors -> the reference runs out of clusters and panics
    with pytest.raises(capi.VersError) as e:
        index.search_approximate(X[0], 41)
    assert e.value.status == capi.ERR_INSUFFICIENT
    assert len(index.search_approximate(X[0], 40)) == 40   # exactly all of them: spills through every list
    # nprobe extension returns what exists instead
    ids, dist, cnt = index.search_batch(X[:2], 50, nprobe=4)
    assert list(cnt) == [40, 40]
    # NaN query -> panic in partial_cmp().unwrap()
    q = X[0].copy(); q[0] = np.nan
    with pytest.raises(capi.VersError) as e:
        index.search_approximate(q, 3)
    assert e.value.status == capi.ERR_NAN
    assert index.search_approximate(X[5], 1)[0] == (5, np.float32(0.0))
    index.close(); empty.close()


def test_add_overflows_slack_and_relayouts():
    d, n, k = 16, 300, 3
    X = dg.dist_c(21, n, d, 3, dg.default_sigma(d))
    index = IVFFlatIndex.build_index(k, 1, 4, X, init_indices=np.array([0, 1, 2], dtype=np.uint64))
    extra = dg.dist_c(22, 400, d, 3, dg.default_sigma(d), seed_c=21 ^ 0xC0FFEE)   # far more than the slack of any list
    for x in extra:
        c, _ = index.add(x)
        assert c == co.add_cluster(index.centroids, x)
    assert index.info()[0] == n + 400
    for q in (extra[7], X[3], dg.dist_u(5, 1, d)[0]):
        for top_k, nprobe in [(10, 0), (64, 2)]:
            ids, dist, cnt = index.search_batch(q, top_k, nprobe)
            oi, od = (co.search_approximate(index.values, index.centroids, index.ids, q, top_k) if nprobe == 0 else
                      co.search_nprobe(index.values, index.centroids, index.ids, q, top_k, nprobe))
            assert np.array_equal(ids[0, :len(oi)], oi) and np.array_equal(bits(dist[0, :len(oi)]), bits(od))
    index.close()


def test_coarse_ahead_is_bit_identical():
    """vers_ivf_coarse_ahead_dev: the next batch's staged queries + ranked lists computed on the side stream must give
    the same bits as the search computing them itself -- alternating slots, a stale slot, a batch never searched."""
    import torch
    from tests.golden import make_golden as mg
    n, d, k, b, top_k, nprobe = 20000, 96, 64, 256, 10, 8
    X = dg.dist_c(0x51, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x51, 1, k, n))
    dev = torch.device("cuda:0")
    Q = torch.from_numpy(dg.dist_c(0x52, 6 * b, d, 4 * k, dg.default_sigma(d))).to(dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(i, ahead_of=None):
        ids = torch.zeros(b, top_k, dtype=torch.int64, device=dev); dst = torch.zeros(b, top_k, dtype=torch.float32, device=dev)
        cnt = torch.zeros(b, dtype=torch.int32, device=dev)
        if ahead_of is not None:
            ix.coarse_ahead_dev(Q[ahead_of * b:].data_ptr(), d, b, nprobe, st)
        ix.search_dev(Q[i * b:].data_ptr(), d, b, top_k, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
        ix.poll(st)
        return ids.cpu().numpy(), dst.cpu().numpy().view(np.uint32), cnt.cpu().numpy()

    want = [run(i) for i in range(6)]
    ix.coarse_ahead_dev(Q.data_ptr(), d, b, nprobe, st)
    got = [run(i, ahead_of=(i + 1) % 6) for i in range(6)]                # pipelined loop: every search finds its slot
    got += [run(3, ahead_of=5), run(5), run(0, ahead_of=0), run(1)]        # out of order / prepared-for-itself / stale slot left behind
    for g, w in zip(got, want + [want[3], want[5], want[0], want[1]]):
        assert all(np.array_equal(a, c) for a, c in zip(g, w))
    ix.add(X[7] * np.float32(1.01), 0)                                      # lists change, centroids do not: slots stay valid
    ix.coarse_ahead_dev(Q[2 * b:].data_ptr(), d, b, nprobe, st)
    a = run(2); ix2 = run(2)
    assert all(np.array_equal(x, y) for x, y in zip(a, ix2))


def test_concurrent_searches_from_threads():
    """`search_approximate(&self)` is a shared borrow in the reference (ivfflat.rs:153): many threads may search one
    index at once.  Every call leases its own workspace (scratch, status words, staging stream), so the threads enqueue
    side by side; every thread must get the serial answers, bit for bit."""
    import threading
    n, d, k = 12000, 64, 32
    X = dg.dist_c(0x61, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x61, 1, k, n))
    Q = dg.dist_c(0x62, 96, d, 4 * k, dg.default_sigma(d))
    want_single = [ix.search_approximate(Q[i], 10) for i in range(Q.shape[0])]
    want_batch = ix.search_batch(Q, 10, 6)
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for i in range(t, Q.shape[0], 8):                     # reference mode, one query per call
                    got = ix.search_approximate(Q[i], 10)
                    assert [g[0] for g in got] == [w[0] for w in want_single[i]]
                    assert np.array_equal(np.array([g[1] for g in got], dtype=np.float32).view(np.uint32),
                                          np.array([w[1] for w in want_single[i]], dtype=np.float32).view(np.uint32))
                ids, dist, cnt = ix.search_batch(Q, 10, 6)           # batched extension, all threads at once
                assert np.array_equal(ids, want_batch[0]) and np.array_equal(cnt, want_batch[2])
                assert np.array_equal(dist.view(np.uint32), want_batch[1].view(np.uint32))
        except Exception as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]

    # device-pointer calls from four threads, each on its OWN stream, all in flight together; one poll per stream
    import torch
    Qd = torch.from_numpy(Q).cuda()
    torch.cuda.synchronize()
    outs = {}

    def dev_worker(t):
        try:
            st = torch.cuda.Stream()
            ids = torch.zeros(6, 96, 10, dtype=torch.int64, device="cuda"); dist = torch.zeros(6, 96, 10, device="cuda")
            cnt = torch.zeros(6, 96, dtype=torch.int32, device="cuda")
            for rep in range(6):   # back to back, no synchronisation in between
                ix.search_dev(Qd.data_ptr(), d, 96, 10, 6, ids[rep].data_ptr(), dist[rep].data_ptr(), cnt[rep].data_ptr(), st.cuda_stream)
            ix.poll(st.cuda_stream)
            outs[t] = (ids.cpu().numpy().astype(np.uint64), dist.cpu().numpy(), cnt.cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=dev_worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    for t in range(4):
        gi, gd, gc = outs[t]
        for rep in range(6):
            assert np.array_equal(gc[rep], want_batch[2]), (t, rep)
            assert np.array_equal(gi[rep], want_batch[0]) and np.array_equal(gd[rep].view(np.uint32), want_batch[1].view(np.uint32)), (t, rep)


def test_batches_in_flight_on_rotating_streams():
    """ONE host thread rotating its batches over three created streams: the calls share the handle's most recent workspace,
    which orders them across the streams (its `done` event), and every batch is the serial answer."""
    import torch
    n, d, k = 20000, 96, 40
    X = dg.dist_c(0x71, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(0x71, 1, k, n))
    Q = dg.dist_c(0x72, 4 * 128, d, 4 * k, dg.default_sigma(d))
    want = [ix.search_batch(Q[i * 128:(i + 1) * 128], 10, 8) for i in range(4)]
    Qd = torch.from_numpy(Q).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    reps = 12
    ids = torch.zeros(reps, 128, 10, dtype=torch.int64, device="cuda"); dist = torch.zeros(reps, 128, 10, device="cuda")
    cnt = torch.zeros(reps, 128, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for r in range(reps):   # back to back, no synchronisation in between
        st = streams[r % 3]
        ix.search_dev(Qd[(r % 4) * 128:].data_ptr(), d, 128, 10, 8, ids[r].data_ptr(), dist[r].data_ptr(), cnt[r].data_ptr(), st.cuda_stream)
    for st in streams:
        ix.poll(st.cuda_stream)
    gi, gd, gc = ids.cpu().numpy().astype(np.uint64), dist.cpu().numpy(), cnt.cpu().numpy()
    for r in range(reps):
        w = want[r % 4]
        assert np.array_equal(gc[r], w[2]) and np.array_equal(gi[r], w[0]) and np.array_equal(gd[r].view(np.uint32), w[1].view(np.uint32)), r


def test_poll_reports_only_the_polled_streams_calls():
    """The status a device-pointer call latches belongs to ITS stream: thread B's poll must not consume (or be handed)
    thread A's INSUFFICIENT.  A: reference mode with top_k beyond the whole index (the reference panics, ivfflat.rs:169);
    B: clean nprobe batches, in flight at the same time on another stream.  Round 2 read and cleared word 0 of EVERY
    workspace in whichever poll came first."""
    import threading
    import torch
    n, d, k = 600, 32, 8
    X = dg.dist_c(0x81, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x81, 1, k, n))
    Q = torch.from_numpy(dg.dist_c(0x82, 64, d, 4 * k, dg.default_sigma(d))).cuda()
    want = ix.search_batch(Q.cpu().numpy(), 10, 4)
    torch.cuda.synchronize()
    res, errors = {}, []
    gate = threading.Barrier(2)

    def worker(name, top_k, nprobe):
        try:
            st = torch.cuda.Stream()
            ids = torch.zeros(64, top_k, dtype=torch.int64, device="cuda"); dist = torch.zeros(64, top_k, device="cuda")
            cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
            polls = []
            for rep in range(8):
                gate.wait()
                for _ in range(3):
                    ix.search_dev(Q.data_ptr(), d, 64, top_k, nprobe, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), st.cuda_stream)
                gate.wait()                                  # both threads have queued their calls before either polls
                try:
                    ix.poll(st.cuda_stream)
                    polls.append(0)
                except capi.VersError as e:                  # the binding raises on a non-zero status
                    polls.append(e.status)
            res[name] = (polls, ids.cpu().numpy().astype(np.uint64), dist.cpu().numpy(), cnt.cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errors.append((name, repr(e)))
            gate.abort()

    ta = threading.Thread(target=worker, args=("A", 700, 0)); tb = threading.Thread(target=worker, args=("B", 10, 4))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors
    assert res["A"][0] == [capi.ERR_INSUFFICIENT] * 8, res["A"][0]
    assert res["B"][0] == [0] * 8, res["B"][0]
    assert np.array_equal(res["B"][1], want[0]) and np.array_equal(res["B"][3], want[2])
    assert np.array_equal(res["B"][2].view(np.uint32), want[1].view(np.uint32))


def test_scan_events_option_times_single_query_scans_only_on_request():
    """vers_set_option("scan_events", v): the event records around the list-scan launch (vers_ivf_scan_times / vers_ivf_last_scan) cost a
    single-query call 5.5-6 us, so by default (2) only batches make them; 1 = every call, 0 = none."""
    n, d, k = 3000, 24, 12
    X = dg.dist_c(0x5EC1, n, d, 6, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 2, X, init_indices=mg.init_draws(0x5EC1, 1, k, n))
    Q = dg.dist_c(0x5EC2, 40, d, 6, dg.default_sigma(d))
    try:
        ix.search_batch(Q, 5, 3); ix.scan_times(reset=True)
        for _ in range(3):
            ix.search_batch(Q[1], 5, 3)                 # default: single queries are not timed
        assert len(ix.scan_times(reset=True)) == 0
        ix.search_batch(Q, 5, 3)                        # ... batches are
        assert len(ix.scan_times(reset=True)) == 1
        capi.set_option("scan_events", 1)
        r1 = ix.search_batch(Q[1], 5, 3)
        assert ix.last_scan()["ms"] > 0.0 and len(ix.scan_times(reset=True)) == 1
        capi.set_option("scan_events", 0)
        r0 = ix.search_batch(Q, 5, 3)
        assert len(ix.scan_times(reset=True)) == 0
        assert np.array_equal(r1[0], r0[0][1:2]) and np.array_equal(bits(r1[1]), bits(r0[1][1:2]))   # timing never touches results
    finally:
        capi.set_option("scan_events", 2)
        ix.close()


def test_round4_entry_points_and_hooks():
    """vers_ivf_search_sharded_dev without an exchange (g == NULL: a single process) == vers_ivf_search_dev; its argument checks;
    the layout / finish-time / build-phase hooks report what the handle holds and ran."""
    import ctypes as C
    import torch
    from vers_amd.dist import VersGather
    n, d, k = 6000, 96, 24
    X = dg.dist_c(0x4A, n, d, 48, dg.default_sigma(d))
    init = mg.init_draws(0x4A, 1, k, n)
    capi.build_phases(reset=True)
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=init)
    ph = capi.build_phases()
    assert ph["total_ms"] > 0 and ph["install_lists_ms"] > 0 and ph["derive_ms"] > 0 and ph["total_ms"] >= ph["install_lists_ms"] + ph["derive_ms"]
    lay = ix.layout_bytes()
    compact = capi.env_option("memory", 0) == 1 or capi.env_option("rowmajor", -1) == 0   # (the suite also runs under VERS_OPTIONS=memory=1)
    assert lay["rows"] >= n * d * 4 and lay["shadow"] * 2 == lay["rows"] and lay["rowmajor"] == (0 if compact else lay["rows"])   # all three copies by default at this size
    b, top_k = 96, 10
    Q = dg.dist_c(0x4B, b, d, 48, dg.default_sigma(d))
    Qd = torch.from_numpy(Q).cuda()
    for nprobe in (0, 6):
        wi, wd, wc = ix.search_batch(Q, top_k, nprobe)
        oi = torch.zeros(b, top_k, dtype=torch.int64, device="cuda"); od = torch.zeros(b, top_k, device="cuda"); oc = torch.zeros(b, dtype=torch.int32, device="cuda")
        ix.search_sharded_dev(None, Qd.data_ptr(), d, b, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        ix.poll()
        assert np.array_equal(oc.cpu().numpy(), wc)
        for q in range(b):
            c = int(wc[q])
            assert np.array_equal(oi.cpu().numpy().astype(np.uint64)[q, :c], wi[q, :c]) and np.array_equal(bits(od.cpu().numpy()[q, :c]), bits(wd[q, :c]))
    assert ix.last_finish_ms() > 0          # (the nprobe batch went through the matrix-core scan and its exact finish)
    # argument checks: top_k = 0, a gather that says another world than the handle's, one without a callback
    with pytest.raises(capi.VersError):
        ix.search_sharded_dev(None, Qd.data_ptr(), d, b, 0, 6, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    g = VersGather(None, 0, 2)
    with pytest.raises(capi.VersError):
        ix.search_sharded_dev(C.cast(C.byref(g), C.c_void_p), Qd.data_ptr(), d, b, top_k, 6, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    ix.close()
