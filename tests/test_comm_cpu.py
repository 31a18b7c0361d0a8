"""vers_comm_t over torch.distributed (vers_amd.dist.TorchComm), world 2 and 3 on CPU with gloo: the five callbacks the
row-sharded build_index calls -- all_gather (counts), send/recv (the chain of running sums and of the cost fold),
broadcast (centroids), all_to_all_v (rows to the owners of their lists) -- are driven through the C function pointers
of the struct, exactly as libvers_hip.so drives them, on host buffers (device=None)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vers_amd.dist import TorchComm
    cm = TorchComm(device=None)
    st = cm.struct
    ok = st.rank == rank and st.world == world
    # all_gather
    mine = np.arange(5, dtype=np.uint32) + 100 * rank
    allv = np.zeros(5 * world, dtype=np.uint32)
    ok &= st.all_gather(None, vp(mine), vp(allv), mine.nbytes) == 0
    ok &= np.array_equal(allv, np.concatenate([np.arange(5, dtype=np.uint32) + 100 * r for r in range(world)]))
    # the chain: rank r continues rank r-1's running sum IN ORDER (f32, non-associative values)
    vals = [np.float32(x) for x in (1e8, 1.0, -1e8, 3.0, 0.25, 7.0)]
    per = len(vals) // world
    lo, hi = rank * per, (len(vals) if rank == world - 1 else (rank + 1) * per)
    acc = np.zeros(1, dtype=np.float32)
    if rank > 0:
        ok &= st.recv(None, vp(acc), 4, rank - 1) == 0
    for v in vals[lo:hi]:
        acc[0] = np.float32(acc[0] + v)
    if rank + 1 < world:
        ok &= st.send(None, vp(acc), 4, rank + 1) == 0
    ok &= st.broadcast(None, vp(acc), 4, world - 1) == 0
    want = np.float32(0)
    for v in vals:
        want = np.float32(want + v)
    ok &= acc[0].view(np.uint32) == np.float32(want).view(np.uint32)
    # all_to_all_v: rank r sends (r + t + 1) words to rank t, value = 1000*r + t
    sb = (C.c_uint64 * world)(); so = (C.c_uint64 * world)(); rb = (C.c_uint64 * world)(); ro = (C.c_uint64 * world)()
    send = []
    off = 0
    for t in range(world):
        n = rank + t + 1 if (rank + t) % 3 != 2 else 0   # some empty pairs
        sb[t] = 4 * n; so[t] = off; off += 4 * n
        send += [1000 * rank + t] * n
    send = np.array(send + [0], dtype=np.uint32)
    off = 0
    for s in range(world):
        n = s + rank + 1 if (s + rank) % 3 != 2 else 0
        rb[s] = 4 * n; ro[s] = off; off += 4 * n
    recv = np.zeros(off // 4 + 1, dtype=np.uint32)
    ok &= st.all_to_all_v(None, vp(send), sb, so, vp(recv), rb, ro) == 0
    exp = []
    for s in range(world):
        exp += [1000 * s + rank] * (int(rb[s]) // 4)
    ok &= np.array_equal(recv[:-1], np.array(exp, dtype=np.uint32))
    ok &= cm.calls["all_to_all_v"] == 1 and cm.calls["all_gather"] == 1
    # vers_gather_t (the sharded search's one exchange) on host buffers: [2][b][top_k] u64 per rank -> [world][2][b][top_k]
    from vers_amd.dist import TorchGather
    tg = TorchGather(device=None)
    part = (np.arange(2 * 3 * 4, dtype=np.uint64) + np.uint64(1000 * rank)).reshape(2, 3, 4)
    gathered = np.zeros((world, 2, 3, 4), dtype=np.uint64)
    g = tg.struct
    ok &= (g.rank, g.world) == (rank, world)
    ok &= g.all_gather_async(None, vp(part), vp(gathered), part.nbytes, None) == 0
    ok &= all(np.array_equal(gathered[r], (np.arange(24, dtype=np.uint64) + np.uint64(1000 * r)).reshape(2, 3, 4)) for r in range(world))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_torch_comm_callbacks_gloo(world):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
