"""world_size-2 (and 3) gloo test of the sharded search protocol on CPU: shard plan (the product's
host-side LPT, vers_shard_plan), the key format, the single all-gather (vers_amd.dist) and the merge
rule.  Device compute is stood in by the oracle HERE IN THE TEST (the product has no CPU path); what
is checked is that plan + exchange + merge reproduce the unsharded reference result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi

KEY_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def order_bits(d):
    b = np.ascontiguousarray(d, dtype=np.float32).view(np.uint32).astype(np.uint64)
    neg = (b >> np.uint64(31)) != 0
    return np.where(neg, (~b) & np.uint64(0xFFFFFFFF), b | np.uint64(0x80000000))


def partial_for_rank(values, centroids, ids, owner, rank, q, top_k, nprobe):
    """What one GPU would emit: keys/ids of its own lists in the query's global probe order."""
    cd = np.array([co.squared_euclidean(c, q) for c in centroids], dtype=np.float32)
    ranked = np.argsort(cd, kind="stable")
    keys = np.full(top_k, KEY_MAX, dtype=np.uint64); out_ids = np.zeros(top_k, dtype=np.uint64)
    if nprobe == 0:  # reference mode: position-wise, list j owns positions [sum take_<j, +take_j)
        rem, pos = top_k, 0
        for c in ranked:
            take = min(rem, len(ids[c]))
            if take and owner[c] == rank:
                dd = np.array([co.squared_euclidean(values[i], q) for i in ids[c]], dtype=np.float32)
                o = np.argsort(dd, kind="stable")[:take]
                keys[pos:pos + take] = (order_bits(dd[o]) << np.uint64(32)) | np.arange(take, dtype=np.uint64)  # seq irrelevant across ranks here
                out_ids[pos:pos + take] = np.asarray(ids[c], dtype=np.uint64)[o]
            pos += take; rem -= take
            if rem == 0:
                break
        return keys, out_ids
    cand = []
    pref = 0
    for c in ranked[:nprobe]:
        if owner[c] == rank:
            for t, i in enumerate(ids[c]):
                dd = co.squared_euclidean(values[i], q)
                cand.append((int(order_bits(np.array([dd]))[0]) << 32 | (pref + t), i))
        pref += len(ids[c])
    cand.sort()
    for j, (kk, i) in enumerate(cand[:top_k]):
        keys[j] = np.uint64(kk); out_ids[j] = np.uint64(i)
    return keys, out_ids


def merge(allp, top_k, nprobe):
    """vers_topk_merge_dev's rule restated in numpy."""
    world, _, b, _ = allp.shape
    out = []
    for q in range(b):
        keys = allp[:, 0, q, :]; ids = allp[:, 1, q, :]
        if nprobe == 0:
            r = np.argmin(keys, axis=0)
            kk = keys[r, np.arange(top_k)]; ii = ids[r, np.arange(top_k)]
        else:
            flat = np.argsort(keys.reshape(-1), kind="stable")[:top_k]
            kk = keys.reshape(-1)[flat]; ii = ids.reshape(-1)[flat]
        ok = kk != KEY_MAX
        out.append(ii[ok])
    return out


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vers_amd.dist import all_gather_partials
    n, d, k = 600, 12, 9
    X = dg.dist_c(3, n, d, 11, dg.default_sigma(d))
    b_ = co.build_index(X, k, 1, 4, mg.init_draws(4, 1, k, n))
    owner = capi.shard_plan(np.array([len(l) for l in b_["ids"]], dtype=np.uint64), world)
    Q = dg.dist_c(5, 6, d, 11, dg.default_sigma(d))
    ok = True
    for nprobe, top_k in [(0, 10), (0, 64), (4, 10), (9, 20)]:
        part = np.zeros((2, Q.shape[0], top_k), dtype=np.uint64)
        for qi, q in enumerate(Q):
            part[0, qi], part[1, qi] = partial_for_rank(X, b_["centroids"], b_["ids"], owner, rank, q, top_k, nprobe)
        allp = all_gather_partials(torch.from_numpy(part.view(np.int64))).numpy().view(np.uint64)
        got = merge(allp, top_k, nprobe)
        for qi, q in enumerate(Q):
            want, _ = (co.search_approximate(X, b_["centroids"], b_["ids"], q, top_k) if nprobe == 0 else
                       co.search_nprobe(X, b_["centroids"], b_["ids"], q, top_k, nprobe))
            ok &= np.array_equal(got[qi], want)
    ret[rank] = bool(ok) and len(set(owner.tolist())) == world
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_protocol_gloo(world):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_plan_is_balanced_and_deterministic():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 5000, size=4096).astype(np.uint64)
    for world in (1, 2, 4, 8):
        o = capi.shard_plan(lens, world)
        assert np.array_equal(o, capi.shard_plan(lens, world))
        load = np.array([lens[o == r].sum() for r in range(world)], dtype=np.float64)
        assert load.max() - load.min() <= lens.max()          # LPT bound
        assert set(np.unique(o)) == set(range(world))
