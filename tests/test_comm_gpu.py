"""TorchComm on DEVICE buffers through the RCCL backend ("nccl"), world 1 on the one GPU of the test box: the raw
pointers the library hands over are wrapped zero-copy (__cuda_array_interface__) and the collectives that exist for a
single rank (all_gather, broadcast, all_to_all_v) run in place on them.  (send / recv need a peer: covered with gloo
in test_comm_cpu.py and test_dist_build_gpu.py.)"""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torch_comm_rccl_world1_device_buffers():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from vers_amd.dist import TorchComm
        cm = TorchComm(device=0)
        st = cm.struct
        a = torch.arange(1000, dtype=torch.float32, device="cuda"); b = torch.zeros(1000, dtype=torch.float32, device="cuda")
        assert st.all_gather(None, a.data_ptr(), b.data_ptr(), 4000) == 0
        assert torch.equal(a, b)
        assert st.broadcast(None, b.data_ptr(), 4000, 0) == 0
        c = torch.zeros(1000, dtype=torch.float32, device="cuda")
        sb = (C.c_uint64 * 1)(4000); so = (C.c_uint64 * 1)(0)
        assert st.all_to_all_v(None, a.data_ptr(), sb, so, c.data_ptr(), sb, so) == 0
        assert torch.equal(a, c)
        # the wrapped tensor aliases the buffer (no copy)
        w = cm._wrap(a.data_ptr(), 4000)
        w.view(torch.float32)[3] = -7.0
        torch.cuda.synchronize()
        assert float(a[3]) == -7.0
    finally:
        dist.destroy_process_group()


def test_libvers_rccl_world1_gather_build_callbacks_and_sharded_search():
    """libvers_rccl.so (include/vers_comm_rccl.h) with a one-rank communicator on the test box's GPU: the library makes its
    own ncclComm_t from a unique id (no torch.distributed anywhere), the search's exchange is ONE ncclAllGather queued on the
    caller's stream (vers_gather_t), the build's five callbacks run on device buffers, and vers_ivf_search_sharded_dev /
    vers_ivf_build_sharded_dev through them equal the plain calls bit for bit.  (Peers: RCCL refuses two ranks on one GPU;
    worlds 2 and 3 run the same entry points over gloo in test_dist_build_gpu.py.)"""
    import torch
    from tests import datagen as dg
    from tests.golden import make_golden as mg
    from vers_amd import rccl
    from vers_amd.index import IVFFlatIndex
    cm = rccl.RcclComm(rccl.RcclComm.unique_id(), 0, 1, 0)
    try:
        g = cm._gather
        assert (g.rank, g.world) == (0, 1)
        side = torch.cuda.Stream()
        a = torch.arange(4096, dtype=torch.float32, device="cuda"); b = torch.zeros(4096, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        assert g.all_gather_async(g.ctx, a.data_ptr(), b.data_ptr(), 4096 * 4, side.cuda_stream) == 0
        side.synchronize()
        assert torch.equal(a, b)
        st = cm._comm
        c = torch.zeros(4096, dtype=torch.float32, device="cuda")
        assert st.all_gather(st.ctx, a.data_ptr(), c.data_ptr(), 4096 * 4) == 0 and torch.equal(a, c)
        assert st.broadcast(st.ctx, c.data_ptr(), 4096 * 4, 0) == 0 and torch.equal(a, c)
        e = torch.zeros(4096, dtype=torch.float32, device="cuda")
        sb = (C.c_uint64 * 1)(4096 * 4); so = (C.c_uint64 * 1)(0)
        assert st.all_to_all_v(st.ctx, a.data_ptr(), sb, so, e.data_ptr(), sb, so) == 0 and torch.equal(a, e)
        # build through the RCCL-backed vers_comm_t (one rank: no callback is needed, the entry point takes it all the same),
        # then the sharded search entry: partial -> ncclAllGather on the stream -> merge
        n, d, k = 3000, 64, 16
        X = dg.dist_c(0xA1, n, d, 40, dg.default_sigma(d))
        init = mg.init_draws(0xA1, 1, k, n)
        whole = IVFFlatIndex.build_index(k, 1, 4, X, init_indices=init)
        ix = IVFFlatIndex(d)
        Xd = torch.from_numpy(X).cuda()
        assert ix.build_sharded_dev(Xd.data_ptr(), n, d, 0, n, k, 1, 4, init, cm, want_assignments=True)
        assert np.array_equal(ix.local_assignments, whole.assignments)
        b_, top_k = 33, 10
        Q = dg.dist_c(0xA2, b_, d, 40, dg.default_sigma(d))
        Qd = torch.from_numpy(Q).cuda()
        for nprobe in (0, 5):
            oi = torch.zeros(b_, top_k, dtype=torch.int64, device="cuda"); od = torch.zeros(b_, top_k, device="cuda")
            oc = torch.zeros(b_, dtype=torch.int32, device="cuda")
            for rep in range(3):  # (several calls in flight on the stream: the gather buffers are the leased workspace's)
                ix.search_sharded_dev(cm.gather_ptr(), Qd.data_ptr(), d, b_, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), side.cuda_stream)
            ix.poll(side.cuda_stream)
            wi, wd, wc = whole.search_batch(Q, top_k, nprobe)
            assert np.array_equal(oc.cpu().numpy(), wc)
            for q in range(b_):
                c_ = int(wc[q])
                assert np.array_equal(oi.cpu().numpy().astype(np.uint64)[q, :c_], wi[q, :c_])
                assert np.array_equal(od.cpu().numpy()[q, :c_].view(np.uint32), wd[q, :c_].view(np.uint32))
        oi = torch.zeros(b_, top_k, dtype=torch.int64, device="cuda"); od = torch.zeros(b_, top_k, device="cuda"); oc = torch.zeros(b_, dtype=torch.int32, device="cuda")
        ix.search_exhaustive_sharded_dev(cm.gather_ptr(), Qd.data_ptr(), d, b_, top_k, 0, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), side.cuda_stream)
        ix.poll(side.cuda_stream)
        wi, wd, wc = whole.search_exhaustive(Q, top_k)
        assert np.array_equal(oi.cpu().numpy().astype(np.uint64), wi) and np.array_equal(od.cpu().numpy().view(np.uint32), wd.view(np.uint32))
        ix.close(); whole.close()
    finally:
        cm.close()


def test_libvers_rccl_adopts_torchs_own_communicator_and_reports_its_versions():
    """vers_rccl_adopt: what a host that already owns an ncclComm_t calls -- here torch's own (ProcessGroupNCCL._comm_ptr), world
    1.  The adopted handle's gather and build callbacks run on it; a wrong device is refused (range, and the device the
    communicator lives on); abort marks the handle dead: every later exchange fails at once instead of hanging."""
    import torch
    import torch.distributed as dist
    from vers_amd import capi, rccl
    v = rccl.versions()
    assert v["built_against"] // 10000 == v["running_on"] // 10000 and "rccl" in v["librccl"].lower(), v
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(8, device="cuda"); dist.all_reduce(t)   # (the communicator exists once a collective has run)
        torch.cuda.synchronize()
        pg = dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda", 0))
        ptr = int(pg._comm_ptr())
        assert ptr != 0
        for bad_dev in (-1, 99):
            with pytest.raises(capi.VersError):
                rccl.RcclComm.adopt(ptr, bad_dev)
        cm = rccl.RcclComm.adopt(ptr, 0)
        assert (cm.rank, cm.world) == (0, 1)
        g = cm._gather
        a = torch.arange(2048, dtype=torch.float32, device="cuda"); b = torch.zeros(2048, dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream(); torch.cuda.synchronize()
        assert g.all_gather_async(g.ctx, a.data_ptr(), b.data_ptr(), 2048 * 4, side.cuda_stream) == 0
        side.synchronize()
        assert torch.equal(a, b)
        st = cm._comm
        c = torch.zeros(2048, dtype=torch.float32, device="cuda")
        assert st.all_gather(st.ctx, a.data_ptr(), c.data_ptr(), 2048 * 4) == 0 and torch.equal(a, c)
        cm.close()     # (adopted: torch's communicator is NOT destroyed with the handle)
        dist.all_reduce(t); torch.cuda.synchronize()
        assert float(t[0]) == 1.0
        # a communicator of the library's own, aborted: later exchanges fail at once
        own = rccl.RcclComm(rccl.RcclComm.unique_id(), 0, 1, 0)
        own.abort()
        g2 = own._gather
        assert g2.all_gather_async(g2.ctx, a.data_ptr(), b.data_ptr(), 2048 * 4, side.cuda_stream) == capi.ERR_COMM
        assert own._comm.all_gather(own._comm.ctx, a.data_ptr(), c.data_ptr(), 2048 * 4) == capi.ERR_COMM
        own.close()
    finally:
        dist.destroy_process_group()


def _rccl_peer_worker(rank, world, port, ret):
    """one process per GPU, the library's own communicator: gathers on several streams, the five build callbacks incl. send /
    recv and the grouped all_to_all_v, destroy with nothing pending"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (only carries the 128-byte id)
    from vers_amd import rccl
    cm = rccl.RcclComm.from_torch(rank)
    g, st = cm._gather, cm._comm
    dev = torch.device("cuda", rank)
    ok = True
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = []
    for i, s_ in enumerate(streams):   # three gathers in flight on three streams of ONE communicator (every rank in the same order)
        a = torch.full((1 << 16,), float(rank * 10 + i), device=dev); b = torch.zeros(world << 16, device=dev)
        torch.cuda.synchronize(dev)
        ok &= g.all_gather_async(g.ctx, a.data_ptr(), b.data_ptr(), 4 << 16, s_.cuda_stream) == 0
        outs.append(b)
    torch.cuda.synchronize(dev)
    for i, b in enumerate(outs):
        ok &= all(float(b[r << 16]) == r * 10 + i and float(b[((r + 1) << 16) - 1]) == r * 10 + i for r in range(world))
    x = torch.full((1024,), float(rank), device=dev)
    if rank + 1 < world: ok &= st.send(st.ctx, x.data_ptr(), 4096, rank + 1) == 0
    if rank > 0:
        y = torch.zeros(1024, device=dev); ok &= st.recv(st.ctx, y.data_ptr(), 4096, rank - 1) == 0 and float(y[5]) == rank - 1
    z = torch.full((1024,), float(rank), device=dev); ok &= st.broadcast(st.ctx, z.data_ptr(), 4096, world - 1) == 0 and float(z[9]) == world - 1
    # all_to_all_v: rank r sends (p + 1) * 256 bytes of value 100 r + p to peer p
    sb = (C.c_uint64 * world)(*[(p + 1) * 256 for p in range(world)]); so = (C.c_uint64 * world)(*[sum((q + 1) * 256 for q in range(p)) for p in range(world)])
    rb = (C.c_uint64 * world)(*[(rank + 1) * 256] * world); ro = (C.c_uint64 * world)(*[p * (rank + 1) * 256 for p in range(world)])
    send = torch.cat([torch.full(((p + 1) * 64,), float(100 * rank + p), device=dev) for p in range(world)]); recv = torch.zeros(world * (rank + 1) * 64, device=dev)
    torch.cuda.synchronize(dev)
    ok &= st.all_to_all_v(st.ctx, send.data_ptr(), sb, so, recv.data_ptr(), rb, ro) == 0
    ok &= all(float(recv[p * (rank + 1) * 64]) == 100 * p + rank for p in range(world))
    cm.close()
    ret[rank] = bool(ok)
    dist.barrier(); dist.destroy_process_group()


def test_libvers_rccl_with_peers_when_the_box_has_two_gpus():
    """First contact of the adapter with world > 1: needs >= 2 GPUs -- skipped on the one-GPU test box, runs wherever
    the driver has a multi-GPU node before any multi-GPU number is quoted."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: RCCL refuses two ranks on one device (scripts/probe/rccl_two_ranks_one_gpu.py)")
    world = min(4, torch.cuda.device_count())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_rccl_peer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
