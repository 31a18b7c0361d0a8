"""One query per call (what Index::search_approximate is behind the shim) at nprobe >= 1: since round 5 its list scan streams the
fp16 shadow (scan1h_kernel) and is finished exactly like a batch's -- pre-selection, certificate, exact re-score, exact re-scan when
the certificate fails.  Host-pointer and device-pointer calls, both scans (vers_set_option("single_shadow")), every certificate
forced to fail, long rows, wide results and tiny indexes: bit for bit the CPU restatement's results."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import numpy as np, torch
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
dev = torch.device("cuda:0")
checked = 0
for seed, n, d, k, metric, top_ks, nprobes in CASES:
    X = dg.dist_c(seed, n, d, 4 * k, dg.default_sigma(d))
    init = mg.init_draws(seed, 1, k, n)
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=init, metric=metric)
    Q = dg.dist_c(seed + 0x100, 12, d, 4 * k, dg.default_sigma(d)); Q[3] = X[min(17, n - 1)]
    qd = torch.from_numpy(Q).to(dev)
    for top_k in top_ks:
        idd = torch.zeros(top_k, dtype=torch.int64, device=dev); dd = torch.zeros(top_k, dtype=torch.float32, device=dev); cd = torch.zeros(1, dtype=torch.int32, device=dev)
        for np_ in nprobes:
            for single_shadow in (1, 0):
                capi.set_option("single_shadow", single_shadow)
                st0 = ix.prescan_stats()
                for qi in range(0, 12, 2):
                    oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, np_, metric=metric)
                    ids, dist, cnt = ix.search_batch(Q[qi], top_k, np_)                      # host pointers: vers_ivf_search, b == 1
                    assert cnt[0] == len(oi), (seed, top_k, np_, qi)
                    assert np.array_equal(ids[0, :len(oi)], oi) and np.array_equal(dist[0, :len(oi)].view(np.uint32), od.view(np.uint32)), (seed, top_k, np_, qi, "host")
                    ix.search_dev(qd[qi].data_ptr(), d, 1, top_k, np_, idd.data_ptr(), dd.data_ptr(), cd.data_ptr(), 0)   # device pointers
                    torch.cuda.synchronize(); ix.poll(0)
                    c = int(cd.item())
                    assert c == len(oi) and np.array_equal(idd.cpu().numpy()[:c].astype(np.uint64), oi), (seed, top_k, np_, qi, "dev")
                    assert np.array_equal(dd.cpu().numpy()[:c].view(np.uint32), od.view(np.uint32)), (seed, top_k, np_, qi, "dev")
                    checked += 1
                st1 = ix.prescan_stats()
                took_shadow = st1["batches"] - st0["batches"]
                for bsz in (2, 3):   # batches below the matrix-core scan's smallest: consecutive single queries on the shadow (single_shadow = 1)
                    ids, dist, cnt = ix.search_batch(Q[5:5 + bsz], top_k, np_)
                    for r in range(bsz):
                        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[5 + r], top_k, np_, metric=metric)
                        assert cnt[r] == len(oi) and np.array_equal(ids[r, :len(oi)], oi) and np.array_equal(dist[r, :len(oi)].view(np.uint32), od.view(np.uint32)), (seed, top_k, np_, bsz, r)
                if single_shadow and ix.shadow_state()["active"] and top_k + 16 <= 64:
                    assert ix.prescan_stats()["batches"] - st1["batches"] == 5
                if single_shadow and ix.shadow_state()["active"] and top_k + 16 <= 64:
                    assert took_shadow == 12, (took_shadow, "the single queries went through the shadow scan")
                    if FORCED: assert st1["fallback_queries"] - st0["fallback_queries"] == 12   # ... and every one was re-scanned exactly
                elif not single_shadow:
                    assert took_shadow == 0
    capi.set_option("single_shadow", 1)
    ix.close()
print("checked", checked)
'''

CASES_MAIN = """CASES = [(0xA1, 9000, 96, 48, 0, (1, 10, 40, 58), (1, 8, 48)), (0xA2, 4000, 300, 32, 1, (10,), (6,)), (0xA3, 1500, 3072, 8, 0, (10,), (3,)),
         (0xA4, 50, 64, 3, 0, (10, 30), (1, 3)), (0xA5, 20000, 128, 16, 0, (10,), (16,))]
"""
CASES_FORCED = "CASES = [(0xB1, 6000, 96, 24, 0, (1, 10, 30), (1, 6)), (0xB2, 3000, 200, 16, 1, (10,), (4,))]\n"


def run_body(cases, env_extra, forced):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", f"FORCED = {forced}\n" + cases + BODY], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "checked" in r.stdout


def test_single_queries_match_oracle_on_both_scans():
    run_body(CASES_MAIN, {}, False)


def test_single_queries_with_every_certificate_forced_to_fail():
    run_body(CASES_FORCED, {"VERS_OPTIONS": "prescan=2"}, True)


def test_single_queries_without_a_shadow():
    run_body(CASES_FORCED, {"VERS_SHADOW": "0"}, False)
