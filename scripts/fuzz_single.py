"""Randomised check of the SINGLE-QUERY path (coarse1_kernel -> scan1_kernel -> ivf_merge_kernel; the generic planner when more
than 64 lists are ranked) through both entry points -- the host-pointer call (vers_ivf_search with one row: results written
straight into the pinned block) and device pointers (vers_ivf_search_dev) -- against the same queries run as a BATCH
(other kernels: matrix-core or ordered-chain group scans) and a sample against the C oracle.  Bit equality.
Development aid: python scripts/fuzz_single.py [seconds [first seed]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t0 = time.time(); n_ok = 0; n_q = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100, 6000)); d = int(rng.choice([3, 8, 33, 64, 130, 300, 513, 768, 1100]))
    k = int(rng.choice([1, 3, 17, 64, 65, 130, 300])); k = min(k, n)
    metric = int(rng.integers(0, 2))
    X = dg.dist_c(seed, n, d, max(2, k // 2), dg.default_sigma(d))
    if rng.random() < 0.3:
        X[n // 2:] = X[: n - n // 2]        # exact ties
    if metric:
        X = (X * (0.5 + (np.arange(n) % 5)[:, None] * 0.375)).astype(np.float32)
    ix = IVFFlatIndex.build_index(k, 1, int(rng.integers(1, 3)), X, init_indices=mg.init_draws(seed, 1, k, n), metric=metric)
    for i in range(int(rng.integers(0, 3))):
        ix.add(X[int(rng.integers(0, n))] * np.float32(1.0 + i / 64.0))
    b = 12
    Q = dg.dist_c(seed + 1, b, d, max(2, k // 2), dg.default_sigma(d)); Q[0] = X[n // 3]
    total = ix.values.shape[0]
    Qd = torch.from_numpy(Q).to(dev)
    idd = torch.zeros(128, dtype=torch.int64, device=dev); dd = torch.zeros(128, dtype=torch.float32, device=dev); cd = torch.zeros(1, dtype=torch.int32, device=dev)
    for top_k in sorted({1, min(10, total), min(64, total), min(100, total)}):
        for nprobe in sorted({0, 1, min(k, 5), min(k, 64), min(k, 90)}):
            try:
                ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
            except capi.VersError as e:   # reference mode with fewer reachable rows than top_k: the single query must fail the same way
                for qi in (0, 5):
                    try:
                        ix.search_batch(Q[qi], top_k, nprobe); raise SystemExit(f"MISMATCH seed {seed}: batch failed ({e.status}), single query did not")
                    except capi.VersError as e1:
                        assert e1.status == e.status, (seed, e.status, e1.status)
                continue
            for qi in range(b):
                c = int(cnt[qi])
                i1, d1, c1 = ix.search_batch(Q[qi], top_k, nprobe)
                ok = c1[0] == c and np.array_equal(i1[0, :c], ids[qi, :c]) and np.array_equal(bits(d1[0, :c]), bits(dist[qi, :c]))
                if qi % 3 == 0:                                                    # the same through device pointers
                    idd.zero_(); dd.zero_(); cd.zero_()
                    ix.search_dev(Qd[qi].data_ptr(), d, 1, top_k, nprobe, idd.data_ptr(), dd.data_ptr(), cd.data_ptr(), st)
                    ix.poll(st)
                    i2 = idd.cpu().numpy().astype(np.uint64); d2 = dd.cpu().numpy(); c2 = int(cd.cpu().numpy()[0])
                    ok = ok and c2 == c and np.array_equal(i2[:c], ids[qi, :c]) and np.array_equal(bits(d2[:c]), bits(dist[qi, :c]))
                if not ok:
                    raise SystemExit(f"MISMATCH seed {seed} n={n} d={d} k={k} metric={metric} top_k={top_k} nprobe={nprobe} query {qi}")
                n_q += 1
            qi = int(rng.integers(0, b))
            oi, od = (co.search_approximate(ix.values, ix.centroids, ix.ids, Q[qi], top_k, metric=metric) if nprobe == 0 else
                      co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe, metric=metric))
            if cnt[qi] != len(oi) or not np.array_equal(ids[qi, :len(oi)], oi) or not np.array_equal(bits(dist[qi, :len(oi)]), bits(od)):
                raise SystemExit(f"MISMATCH vs oracle seed {seed} top_k={top_k} nprobe={nprobe} query {qi}")
    ix.close()
    n_ok += 1; seed += 1
print(f"fuzz_single: {n_ok} random configurations, {n_q} single queries == the same queries in a batch == oracle sample, bit for bit")
