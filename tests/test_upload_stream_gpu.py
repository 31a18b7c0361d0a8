"""Streamed rebuild of the device cache (vers_ivf_upload_begin / _chunk / _chunk_dev / _end): the reference's
save -> load -> search sequence (utils.rs:140-148, base.rs:45-58) for an index that no single GPU holds.  The five fields
(ivfflat.rs:8-15) arrive in ragged chunks -- host rows and device rows mixed -- whole and sharded over worlds 2 and 4; every
handle's lists (rows, vec ids, order), its search results and the merged sharded results must equal, bit for bit, what
vers_ivf_upload of the same fields gives and what build_index made; added rows (ivfflat.rs:200-213) travel too."""
import numpy as np
import pytest

from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _stream(ix, X, A, k, cent, cuts, dev_every=0, ld=None):
    """begin -> chunks at `cuts` (every dev_every-th chunk from device memory, with junk in its padding) -> end"""
    import torch
    n, d = X.shape
    ix.upload_begin(cent, np.bincount(A.astype(np.int64), minlength=k).astype(np.uint64), n)
    edges = [0] + list(cuts) + [n]
    for i in range(len(edges) - 1):
        a, b = edges[i], edges[i + 1]
        if dev_every and i % dev_every == 0:
            ldp = ld or d
            Xp = np.full((b - a, ldp), np.nan, dtype=np.float32); Xp[:, :d] = X[a:b]
            Xd = torch.from_numpy(Xp).cuda(); Ad = torch.from_numpy(A[a:b].astype(np.int64)).cuda()
            ix.upload_chunk_dev(Xd.data_ptr(), ldp, Ad.data_ptr(), a, b - a)
        else:
            ix.upload_chunk(X[a:b], A[a:b], a)
    ix.upload_end()


@pytest.mark.parametrize("d,ld", [(50, 56), (768, 768)])
def test_streamed_upload_equals_upload_and_build(d, ld, monkeypatch):
    import torch
    n, k = (2600, 20) if d == 50 else (5000, 16)
    X = dg.dist_c(0x71, n, d, 25, dg.default_sigma(d))
    init = mg.init_draws(6, 1, k, n)
    whole = IVFFlatIndex.build_index(k, 1, 5, X, init_indices=init)
    for x in dg.dist_u(78, 3, d):  # rows appended by add: vec ids n, n+1, n+2 at the end of their lists
        whole.add(x)
    Xall = np.ascontiguousarray(whole.values, dtype=np.float32); A = np.asarray(whole.assignments, dtype=np.uint64)
    n_all = Xall.shape[0]
    ref = IVFFlatIndex(d)  # what vers_ivf_upload makes of the same fields
    ref.values, ref.centroids, ref.assignments, ref.num_centroids = Xall, whole.centroids, A, k
    ref._upload()
    b, top_k = 19, 10
    Q = dg.dist_c(0x72, b, d, 25, dg.default_sigma(d)); Q[2] = Xall[n_all - 1]
    Qd = torch.from_numpy(Q).cuda()
    cuts = [1, 64, 65, 700, 701, 1999, n_all - 1]  # ragged: one row, across tile boundaries, the last row alone
    for world in (1, 2, 4):
        shards = []
        for r in range(world):
            ix = IVFFlatIndex(d)
            if world > 1:
                ix.set_shard(r, world)
            _stream(ix, Xall, A, k, whole.centroids, cuts, dev_every=2 if r % 2 == 0 else 0, ld=ld)
            assert ix.info()[0] == n_all and np.array_equal(ix.list_lengths(), whole.list_lengths())
            shards.append(ix)
        owners = shards[0].owners()
        for c in range(k):
            rows, ids = shards[int(owners[c])].get_list(c)
            r_rows, r_ids = ref.get_list(c)
            assert np.array_equal(ids, np.asarray(whole.ids[c], dtype=np.uint64)) and np.array_equal(ids, r_ids)
            assert np.array_equal(bits(rows), bits(r_rows)) and np.array_equal(bits(rows), bits(Xall[ids.astype(np.int64)]))
        for nprobe in (0, 5):
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
            for src in (whole, ref):
                wi, wd, wc = src.search_batch(Q, top_k, nprobe)
                assert np.array_equal(oc.cpu().numpy(), wc)
                for q in range(b):
                    c = int(wc[q])
                    assert np.array_equal(oi.cpu().numpy().astype(np.uint64)[q, :c], wi[q, :c]) and np.array_equal(bits(od.cpu().numpy()[q, :c]), bits(wd[q, :c]))
        for ix in shards:
            ix.close()
    # a streamed handle is a complete index: add and batched search keep working on it
    ix = IVFFlatIndex(d)
    _stream(ix, Xall, A, k, whole.centroids, cuts, dev_every=3, ld=ld)
    x = dg.dist_u(79, 1, d)[0]
    c = capi.C.c_uint64(0); v = capi.C.c_uint64(0)
    capi.check(capi.lib().vers_ivf_add(ix._h, capi._ptr(np.ascontiguousarray(x)), capi.C.byref(c), capi.C.byref(v)))
    assert (c.value, v.value) == whole.add(x)
    gi, gd, gc = ix.search_batch(Q, top_k, 4); wi, wd, wc = whole.search_batch(Q, top_k, 4)
    assert np.array_equal(gi, wi) and np.array_equal(bits(gd), bits(wd)) and np.array_equal(gc, wc)
    ix.close(); ref.close(); whole.close()


def test_streamed_upload_small_staging_buffer(monkeypatch):
    """the host path's pinned buffer holds fewer rows than a chunk: sub-chunks, same lists"""
    import subprocess, sys, os
    code = (
        "import numpy as np\n"
        "from tests import datagen as dg\n"
        "from tests.golden import make_golden as mg\n"
        "from vers_amd.index import IVFFlatIndex\n"
        "n, d, k = 3000, 40, 12\n"
        "X = dg.dist_c(0x73, n, d, 20, dg.default_sigma(d)); init = mg.init_draws(3, 1, k, n)\n"
        "w = IVFFlatIndex.build_index(k, 1, 4, X, init_indices=init)\n"
        "A = np.asarray(w.assignments, dtype=np.uint64)\n"
        "for world, r in ((1, 0), (3, 1)):\n"
        "    ix = IVFFlatIndex(d)\n"
        "    if world > 1: ix.set_shard(r, world)\n"
        "    ix.upload_begin(w.centroids, np.bincount(A.astype(np.int64), minlength=k).astype(np.uint64), n)\n"
        "    ix.upload_chunk(X[:2000], A[:2000], 0); ix.upload_chunk(X[2000:], A[2000:], 2000); ix.upload_end()\n"
        "    own = ix.owners()\n"
        "    for c in range(k):\n"
        "        if own[c] != r: continue\n"
        "        rows, ids = ix.get_list(c)\n"
        "        assert np.array_equal(ids, np.asarray(w.ids[c], dtype=np.uint64)) and np.array_equal(rows.view(np.uint32), X[ids.astype(np.int64)].view(np.uint32))\n"
        "print('ok')\n")
    env = dict(os.environ, VERS_OPTIONS="upload_stage_mb=0")  # 0 MB -> the floor of 64 rows per sub-chunk
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_streamed_upload_refuses_inconsistent_fields():
    n, d, k = 600, 24, 6
    X = dg.dist_u(0x74, n, d)
    A = (np.arange(n) % k).astype(np.uint64)
    cent = dg.dist_u(0x75, k, d)
    lens = np.bincount(A.astype(np.int64), minlength=k).astype(np.uint64)
    ix = IVFFlatIndex(d)
    with pytest.raises(capi.VersError):  # no upload in progress
        ix.upload_chunk(X[:10], A[:10], 0)
    with pytest.raises(capi.VersError):  # lengths do not add up
        ix.upload_begin(cent, lens, n + 1)
    ix.upload_begin(cent, lens, n)
    with pytest.raises(capi.VersError):  # not contiguous
        ix.upload_chunk(X[10:20], A[10:20], 10)
    ix.upload_chunk(X[:300], A[:300], 0)
    with pytest.raises(capi.VersError):  # rows missing
        ix.upload_end()
    with pytest.raises(capi.VersError):  # the failed _end abandoned the upload
        ix.upload_chunk(X[300:], A[300:], 300)
    # the handle holds no index: the reference's search on an index without centroids is out of bounds (ivfflat.rs:169)
    with pytest.raises(capi.VersError):
        ix.search_batch(X[:2], 3, 0)
    # a list that receives more rows than announced
    ix.upload_begin(cent, lens, n)
    A2 = A.copy(); A2[5] = (A2[5] + 1) % k
    with pytest.raises(capi.VersError):
        ix.upload_chunk(X, A2, 0)
        ix.upload_end()
    # an assignment out of range
    ix.upload_begin(cent, lens, n)
    A3 = A.copy(); A3[7] = k
    with pytest.raises(capi.VersError):
        ix.upload_chunk(X, A3, 0)
    # and after all that a correct sequence still works; the empty index too
    ix.upload_begin(cent, lens, n); ix.upload_chunk(X, A, 0); ix.upload_end()
    assert ix.info()[0] == n
    for c in range(k):  # (the announced assignments, not the nearest centroids, decide the lists: ids[c] is implied by `assignments`)
        rows, ids = ix.get_list(c)
        assert np.array_equal(ids, np.arange(c, n, k, dtype=np.uint64)) and np.array_equal(bits(rows), bits(X[c::k]))
    ix.upload_begin(np.zeros((0, d), np.float32), np.zeros(0, np.uint64), 0); ix.upload_end()
    assert ix.info()[:2] == (0, 0)
    ix.close()


def test_kmeans_assign_dev_equals_host_entry():
    """vers_kmeans_assign_dev (what a host streaming a corpus through a trained quantiser calls per chunk) == vers_kmeans_assign:
    small (exact scan) and large enough for the matrix-core path, junk in the rows' padding columns."""
    import torch
    for n, d, k, ld in ((700, 30, 9, 32), (40000, 64, 512, 72)):
        X = dg.dist_c(0x76, n, d, 40, dg.default_sigma(d)); Cn = dg.dist_u(0x77, k, d)
        a_ref, m_ref = capi.kmeans_assign(X, Cn, want_min_dist=True)
        Xp = np.full((n, ld), np.nan, dtype=np.float32); Xp[:, :d] = X
        Xd = torch.from_numpy(Xp).cuda(); Cd = torch.from_numpy(Cn).cuda()
        out = torch.zeros(n, dtype=torch.int64, device="cuda"); md = torch.zeros(n, dtype=torch.float32, device="cuda")
        capi.kmeans_assign_dev(Xd.data_ptr(), n, ld, Cd.data_ptr(), k, d, d, out.data_ptr(), md.data_ptr())
        assert np.array_equal(out.cpu().numpy().astype(np.uint64), a_ref) and np.array_equal(bits(md.cpu().numpy()), bits(m_ref))
    # the scratch is kept per DEVICE between calls (also when another thread calls) and a call with n == 0 releases it
    import threading
    held, _ = capi.mem_stats()
    t = threading.Thread(target=lambda: capi.kmeans_assign_dev(Xd.data_ptr(), n, ld, Cd.data_ptr(), k, d, d, out.data_ptr(), md.data_ptr()))
    t.start(); t.join()
    assert capi.mem_stats()[0] == held, "a second calling thread must reuse the device's scratch, not allocate its own"
    assert np.array_equal(out.cpu().numpy().astype(np.uint64), a_ref)
    capi.kmeans_assign_release()
    assert capi.mem_stats()[0] < held
    capi.kmeans_assign_release()   # (nothing held: a no-op)
