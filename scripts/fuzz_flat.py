"""Randomised check of the flat index's SINGLE-QUERY paths -- the fp16 shadow (flat1h_kernel + exact finish) and, with
vers_set_option("single_shadow", 0), the f32 ordered chains -- against each other and the C oracle: random n (tile remainders,
tiny corpora), d (padding columns), top_k, metric, exact ties, rows of very different norms.  Bit equality.
Development aid: python scripts/fuzz_flat.py [seconds [first seed]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from vers_amd import capi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
t0 = time.time(); n_cfg = 0; n_q = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 4097, 20000, 70001])); d = int(rng.choice([1, 3, 8, 33, 64, 100, 128, 300, 513, 768, 1500]))
    if n * d > 12_000_000: n = max(1, 12_000_000 // d)
    metric = int(rng.integers(0, 2))
    X = dg.dist_c(seed, n, d, max(2, min(n, 40)), dg.default_sigma(d)) if rng.random() < 0.5 else dg.dist_u(seed, n, d)
    if rng.random() < 0.3 and n > 1:
        X[n // 2:] = X[: n - n // 2]          # exact ties: the lower index first (utils.rs:77)
    if rng.random() < 0.3:
        X = (X * (0.25 + (np.arange(n) % 7)[:, None] * 0.5)).astype(np.float32)
    Q = dg.dist_u(seed + 1, 5, d); Q[0] = X[n // 3]
    fc = capi.FlatCorpus(d); fc.upload(X)
    for top_k in sorted(set(int(x) for x in rng.choice([1, 2, 10, 33, 58, 59, 64], size=3))):
        res = {}
        for ssh in (1, 0):
            capi.set_option("single_shadow", ssh)
            res[ssh] = [fc.search(Q[qi], top_k, metric) for qi in range(5)]
        for qi in range(5):
            (i1, d1, c1), (i0, d0, c0) = res[1][qi], res[0][qi]
            assert c1[0] == c0[0] and np.array_equal(i1, i0) and np.array_equal(bits(d1), bits(d0)), ("shadow vs f32", seed, n, d, metric, top_k, qi)
            if qi < 2:
                oi, od = co.search_exhaustive(X, Q[qi], top_k, metric=metric)
                assert c1[0] == len(oi) and np.array_equal(i1[0, :len(oi)], oi) and np.array_equal(bits(d1[0, :len(oi)]), bits(od)), ("vs oracle", seed, n, d, metric, top_k, qi)
            n_q += 1
    capi.set_option("single_shadow", 1)
    fc.close()
    n_cfg += 1; seed += 1
print(f"fuzz_flat: {n_cfg} random configurations, {n_q} single queries: shadow == f32 chains == oracle sample, bit for bit")
