"""build_index with the assign step sharded across processes (vers_ivf_set_build_shard): two and three processes
share the one GPU of the test box and exchange through gloo (RCCL on a real node, same code path around it); every
process must end with the single-process result -- centroid bits, assignments, cost bits -- which is the oracle's."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def worker(rank, world, port, force_mfma, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if force_mfma:
        os.environ["VERS_ASSIGN"] = "2"
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import datagen as dg
    from tests.golden import make_golden as mg
    from vers_amd.index import IVFFlatIndex
    n, d, k = 5003, 96, 70            # n is not a multiple of the 64-row chunk granularity: ragged last chunk
    X = dg.dist_c(0xD1, n, d, 210, dg.default_sigma(d))
    init = mg.init_draws(0xD1, 2, k, n)
    ix = IVFFlatIndex.build_index(k, 2, 5, X, init_indices=init, device=0, build_shard=(rank, world))
    q = dg.dist_c(0xD2, 40, d, 210, dg.default_sigma(d))
    ids, dist_, cnt = ix.search_batch(q, 10, 6)
    ret[rank] = (np.ascontiguousarray(ix.centroids).view(np.uint32).copy(), np.asarray(ix.assignments).copy(),
                 np.float32(ix.cost).view(np.uint32), ids.copy(), dist_.view(np.uint32).copy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,force_mfma", [(2, False), (3, True)])
def test_sharded_assign_build_is_bit_exact(world, force_mfma):
    from oracle import c_oracle as co
    from tests import datagen as dg
    from tests.golden import make_golden as mg
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), force_mfma, ret), nprocs=world, join=True)
    n, d, k = 5003, 96, 70
    X = dg.dist_c(0xD1, n, d, 210, dg.default_sigma(d))
    o = co.build_index(X, k, 2, 5, mg.init_draws(0xD1, 2, k, n))
    for r in range(world):
        cb, a, cost_bits, ids, db = ret[r]
        assert np.array_equal(cb, np.ascontiguousarray(o["centroids"]).view(np.uint32)), r
        assert np.array_equal(a, o["assignments"]), r
        assert cost_bits == np.float32(o["cost"]).view(np.uint32), r
        assert np.array_equal(ids, ret[0][3]) and np.array_equal(db, ret[0][4])
