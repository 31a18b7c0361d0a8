"""Batched list scan on the matrix cores (csrc/prescan.hip.h; fp16 shadow rows by default, f32 rows with VERS_SHADOW=0):
MFMA pre-selection + exact re-score + certificate
+ exact re-scan of uncertified queries must return the SAME bits as the ordered-chain scan and the oracle --
ids, order, distance bits -- on clustered, uniform and heavily tied data, ragged lists, after add(), and with the
certificate forced to fail."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import os
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi, testhooks
from vers_amd.index import IVFFlatIndex

def check(ix, Q, top_k, nprobe, step=7):
    ids, dist, cnt = ix.search_batch(Q, top_k, nprobe)
    for qi in range(0, Q.shape[0], step):
        oi, od = co.search_nprobe(ix.values, ix.centroids, ix.ids, Q[qi], top_k, nprobe)
        assert cnt[qi] == len(oi), (nprobe, top_k, qi, cnt[qi], len(oi))
        assert np.array_equal(ids[qi, :len(oi)], oi), (nprobe, top_k, qi, ids[qi, :len(oi)], oi)
        assert np.array_equal(dist[qi, :len(oi)].view(np.uint32), od.view(np.uint32)), (nprobe, top_k, qi)
        assert not ids[qi, len(oi):].any() and not dist[qi, len(oi):].any()   # past the count: zeros, not stale memory

total = 0
# (1) clustered data, several shapes (d = 300 pads to 320 columns; lists are ragged: lengths not multiples of 64)
for seed, n, d, k, b, nprobe, top_ks in [(0x81, 9000, 96, 48, 160, 8, (1, 10, 26, 48)), (0x82, 5000, 300, 32, 96, 6, (10,)),
                                         (0x83, 3000, 768, 24, 80, 5, (10, 20)),
                                         # d = 1152: the 32-query block just fits LDS; d = 1536: it does not -- the NARROW variant's 16 queries do
                                         # (round 5, fp16 shadow: 32 queries with the query block as fp16 hi ONLY up to d = 2304, 16 up to d = 4608)
                                         (0x84, 1500, 1152, 12, 64, 4, (10,)), (0x85, 1200, 1536, 12, 64, 4, (10,)), (0x87, 900, 2304, 8, 40, 3, (10,)),
                                         (0x88, 700, 3072, 6, 36, 3, (10,)), (0x89, 500, 4608, 5, 24, 2, (10,)),
                                         # a small batch over many lists: < 2 queries per list (round 3: one ordered-chain scan per (query, list) pair)
                                         (0x86, 6000, 64, 96, 12, 6, (10, 30))]:
    X = dg.dist_c(seed, n, d, 4 * k, dg.default_sigma(d))
    ix = IVFFlatIndex.build_index(k, 1, 3, X, init_indices=mg.init_draws(seed, 1, k, n))
    Q = dg.dist_c(seed + 0x100, b, d, 4 * k, dg.default_sigma(d)); Q[3] = X[17]
    b0 = ix.prescan_stats()["batches"]
    for top_k in top_ks:
        check(ix, Q, top_k, nprobe)
        total += 1
    in_domain = d <= 2304 or capi.env_option("shadow", 1) != 0   # (f32 rows: d > 2304 stays on the ordered chains)
    if capi.env_option("prescan", 1) != 0 and in_domain:   # every one of these shapes is inside the matrix-core scan's domain
        assert ix.prescan_stats()["batches"] - b0 == len(top_ks), (seed, d, b, ix.prescan_stats()["batches"] - b0)
    if seed == 0x81:
        for i in range(70):  # add(): the new rows' norms are maintained incrementally; one list outgrows its slack
            ix.add(Q[i % 5] * np.float32(1.0 + i / 512.0), 0)
        check(ix, Q, 10, nprobe)
        total += 1
        check(ix, Q, 40, nprobe)   # wide lists (50 keys)
        check(ix, Q, 54, nprobe)   # top_k + 16 keys of slack > 64: stays on the ordered-chain scan (round 6: a slack of 6 .. 15 failed the certificate too often)
        check(ix, Q, 60, nprobe)
# (2) uniform data: distances concentrate, many near-ties around the k-th
X = dg.dist_u(0x91, 6000, 64)
ix = IVFFlatIndex.build_index(40, 1, 2, X, init_indices=mg.init_draws(0x91, 1, 40, 6000))
check(ix, dg.dist_u(0x92, 128, 64), 10, 8)
total += 1
# (3) heavy ties: every vector stored 40 times -> exact ties far denser than the slack; seq order must decide
B = dg.dist_c(0xA1, 150, 32, 30, dg.default_sigma(32))
X = np.repeat(B, 40, axis=0)
ix = IVFFlatIndex.build_index(16, 1, 2, X, init_indices=mg.init_draws(0xA1, 1, 16, X.shape[0]))
Q = dg.dist_c(0xA2, 64, 32, 30, dg.default_sigma(32)); Q[0] = B[5]
check(ix, Q, 10, 6, step=3)
total += 1
st = ix.prescan_stats()
print("TIES", st["batches"], st["fallback_queries"])
# ... the fp16 shadow (the default) keeps top_k + 24 keys: 100 copies of every vector defeat its certificate on every query,
# and then the failure watch must switch it off for the handle (once 1/8 of >= 256 queries had to be re-scanned exactly)
if capi.env_option("shadow", 1) != 0 and capi.env_option("prescan", 1) == 1:
    X2 = np.repeat(B[:60], 100, axis=0)
    ix2 = IVFFlatIndex.build_index(8, 1, 2, X2, init_indices=mg.init_draws(0xA3, 1, 8, X2.shape[0]))
    assert ix2.shadow_state()["active"]
    for rep in range(6):
        check(ix2, Q, 10, 4, step=9)
    assert not ix2.shadow_state()["active"], "the failure watch did not switch the shadow off"
    check(ix2, Q, 10, 4, step=5)
# (3a) ONE query of the batch fails (it equals a vector stored 70 times, the others see no ties): its exact re-scan is
#      spread over a whole group of blocks, each list cut into chunks of tiles
X = dg.dist_c(0xD1, 6000, 96, 40, dg.default_sigma(96)); X[100:170] = X[100]
ix = IVFFlatIndex.build_index(10, 1, 2, X, init_indices=mg.init_draws(0xD1, 1, 10, 6000))
Qs = dg.dist_c(0xD2, 48, 96, 40, dg.default_sigma(96)); Qs[5] = X[100]
f0 = ix.prescan_stats()["fallback_queries"]
check(ix, Qs, 10, 6, step=1)
print("ONE", ix.prescan_stats()["fallback_queries"] - f0)
# ... and TEN of them (ten different vectors stored 70 times each): groups of 8 blocks, two chunks of tiles per list
for v in range(10):
    X[1000 + 100 * v:1070 + 100 * v] = X[1000 + 100 * v]
ix = IVFFlatIndex.build_index(10, 1, 2, X, init_indices=mg.init_draws(0xD1, 1, 10, 6000))
Qs = dg.dist_c(0xD2, 48, 96, 40, dg.default_sigma(96))
for v in range(10):
    Qs[3 + 4 * v] = X[1000 + 100 * v]
f0 = ix.prescan_stats()["fallback_queries"]
check(ix, Qs, 10, 6, step=1)
print("TEN", ix.prescan_stats()["fallback_queries"] - f0)
# (3b) duplicated rows leave k-means clusters empty (zero centroids at distance |q|^2 = 1, nearer than other modes'
#      centroids): most queries probe nothing but empty lists and must come back with count 0
X = dg.dist_c(0xC1, 4500, 130, 72, dg.default_sigma(130)); X[2250:] = X[:2250]
ix = IVFFlatIndex.build_index(72, 1, 2, X, init_indices=mg.init_draws(0xC1, 1, 72, 4500))
Qe = dg.dist_c(0xC2, 140, 130, 72, dg.default_sigma(130))
check(ix, Qe, 10, 3, step=1)
print("EMPTY", int((ix.search_batch(Qe, 10, 3)[2] == 0).sum()))
# (4) large magnitudes: |x|^2 overflows while the distances stay finite -> nothing is finite on the matrix cores,
#     every query is re-scanned exactly
big = lambda s, n: (np.float32(1.5e19) * (np.float32(1.0) + np.float32(1e-3) * dg.dist_u(s, n, 16))).astype(np.float32)
X = big(0xB1, 2000)
ix = IVFFlatIndex.build_index(8, 1, 2, X, init_indices=mg.init_draws(0xB1, 1, 8, 2000))
check(ix, big(0xB2, 64), 5, 4, step=5)
st = ix.prescan_stats()
print("HUGE", st["batches"], st["fallback_queries"])
# (5) tiny magnitudes: the elements sit in fp16's subnormal range, the shadow keeps one or two bits of them -- the measured
#     residual makes the certificate fail instead of trusting it, and the results stay the oracle's
tiny = lambda s, n: (np.float32(3e-7) * dg.dist_c(s, n, 48, 24, dg.default_sigma(48))).astype(np.float32)
X = tiny(0xE1, 3000)
ix = IVFFlatIndex.build_index(12, 1, 2, X, init_indices=mg.init_draws(0xE1, 1, 12, 3000))
check(ix, tiny(0xE2, 72), 10, 5, step=4)
# (6) large but representable magnitudes (|x| ~ 300): nothing special happens
X = (np.float32(300.0) * dg.dist_c(0xE3, 3000, 48, 24, dg.default_sigma(48))).astype(np.float32)
ix = IVFFlatIndex.build_index(12, 1, 2, X, init_indices=mg.init_draws(0xE3, 1, 12, 3000))
f0 = ix.prescan_stats()["fallback_queries"]
check(ix, (np.float32(300.0) * dg.dist_c(0xE4, 72, 48, 24, dg.default_sigma(48))).astype(np.float32), 10, 5, step=4)
print("BIG", ix.prescan_stats()["fallback_queries"] - f0)
# (7) what the rows that hold no vector contain (slack behind the lists, tile padding: uninitialised memory) must not matter:
#     inf / NaN there once reached the real rows of the same tile through a 0-weighted term of the |x|^2 MFMA of the fp16 scan
Qb = (np.float32(300.0) * dg.dist_c(0xE4, 72, 48, 24, dg.default_sigma(48))).astype(np.float32)
worst = 0
for v in (float("inf"), float("nan"), -1.0e30, 1.5e19):
    testhooks.poison_slack(ix, v)
    f0 = ix.prescan_stats()["fallback_queries"]
    check(ix, Qb, 10, 5, step=4)
    worst = max(worst, ix.prescan_stats()["fallback_queries"] - f0)
print("SLACK", worst)
print("TOTAL", total)
'''


def run(env_extra):
    env = dict(os.environ); env.update(env_extra); env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", BODY], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return {l.split()[0]: tuple(int(v) for v in l.split()[1:]) for l in r.stdout.splitlines() if l[:4] in ("TIES", "HUGE", "TOTA") or l[:3] in ("ONE", "BIG", "TEN") or l[:5] == "SLACK"}


def test_matrix_core_list_scan_is_bit_exact():
    out = run({})                                               # the default: fp16 shadow rows feed the pre-selection
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0          # ties denser than the slack fail the certificate ...
    assert out["HUGE"] == (1, 64)                               # ... and so does every query whose values overflow
    assert out["ONE"] == (1,) and out["BIG"] == (0,) and out["TEN"][0] >= 10 and out["SLACK"] == (0,)


def test_f32_rows_feed_the_scan_without_the_shadow():
    out = run({"VERS_SHADOW": "0"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10 and out["BIG"] == (0,) and out["SLACK"] == (0,)


def test_exact_finish_gathers_from_the_tiles_without_the_row_major_copy():
    """VERS_ROWMAJOR=0: the survivors' rows come out of the lane-transposed tiles (16-byte pieces 1 KB apart) instead of the
    row-major second copy: same staged chains (csrc/staged.hip.h), another address pattern."""
    out = run({"VERS_ROWMAJOR": "0"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10 and out["BIG"] == (0,) and out["SLACK"] == (0,)


def test_forced_certificate_failure_is_exact():
    out = run({"VERS_OPTIONS": "prescan=2"})
    assert out["TIES"] == (1, 64)
    out = run({"VERS_OPTIONS": "prescan=2", "VERS_SHADOW": "0"})
    assert out["TIES"] == (1, 64)


def test_ordered_chain_scan_still_available():
    out = run({"VERS_OPTIONS": "prescan=0"})
    assert out["TIES"] == (0, 0) and out["HUGE"] == (0, 0)


def test_results_do_not_depend_on_uninitialised_memory():
    """The same body with NaN in every storage row that holds no vector and 0x7f in every new device buffer of the library
    (diagnosis knobs of DESIGN.md section 5): the oracle comparisons inside hold and the certificate statistics do not move."""
    out = run({"VERS_OPTIONS": "poison_slack_bits=0x7fc00000,poison_alloc=0x7f"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64)
    assert out["ONE"] == (1,) and out["BIG"] == (0,) and out["TEN"][0] >= 10 and out["SLACK"] == (0,)


def test_narrow_query_blocks_are_bit_exact():
    """option pre_narrow=1: 16 queries per block at every d (the variant that d = 1536 .. 2304 need), on the fp16 shadow and on the
    f32 rows, and with every certificate forced to fail."""
    out = run({"VERS_OPTIONS": "pre_narrow=1"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10 and out["SLACK"] == (0,)
    out = run({"VERS_OPTIONS": "pre_narrow=1", "VERS_SHADOW": "0"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["BIG"] == (0,)
    out = run({"VERS_OPTIONS": "pre_narrow=1,prescan=2"})
    assert out["TIES"] == (1, 64)


def test_hi_only_query_blocks_are_bit_exact():
    """option pre_hi_only=1: the fp16 shadow scan with the query block as fp16 hi only (prescan_kernel_g<.., LO = false>: what
    1152 < d <= 2304 take with 32 queries per block and d <= 4608 with 16) at EVERY d, wide and narrow blocks, and with every
    certificate forced to fail; the certificate charges the query's measured fp16 residual."""
    out = run({"VERS_OPTIONS": "pre_hi_only=1"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10 and out["SLACK"] == (0,)
    out = run({"VERS_OPTIONS": "pre_hi_only=1,pre_narrow=1"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10
    out = run({"VERS_OPTIONS": "pre_hi_only=1,prescan=2"})
    assert out["TIES"] == (1, 64)


def test_32_query_hi_lo_blocks_without_the_wide_variant():
    """option pre_wide=0: the 32-query blocks with both halves of the query's fp16 split (what every d <= 1152 ran until round 5 and
    960 < d <= 1152 still runs); the default covers the 64-query hi-only blocks at small d."""
    out = run({"VERS_OPTIONS": "pre_wide=0"})
    assert out["TIES"][0] == 1 and out["TIES"][1] > 0 and out["HUGE"] == (1, 64) and out["ONE"] == (1,) and out["TEN"][0] >= 10 and out["BIG"] == (0,) and out["SLACK"] == (0,)
    out = run({"VERS_OPTIONS": "pre_wide=0,prescan=2"})
    assert out["TIES"] == (1, 64)
