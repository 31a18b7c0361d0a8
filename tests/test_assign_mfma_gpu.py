"""k-means assign through the f32 matrix cores (km_assign_mfma: GEMM pre-selection + exact re-score of the
candidate + certificate + exact scan for the uncertified points) must give the SAME bits as the exact scan and
the oracle: assignments, minimum distances, and -- through build_index -- centroids and cost."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from tests.golden import make_golden as mg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

# (1) primitive, several shapes: in-place rows (d == 64: pitch == padded length), staged rows, ragged tails,
#     duplicate centroids (ties -> lower index), points equal to a centroid, k not a multiple of 128; the last two: shapes of the
#     fp16 x fp16 contraction staged by LDS-DMA (256-centroid tiles, rows of whole 128-column groups; d = 200: padding columns)
for n, d, k, kind in [(5000, 64, 130, "u"), (4097, 96, 300, "c"), (777, 20, 3, "u"), (300, 300, 64, "c"), (129, 768, 257, "c"), (9, 5, 20, "u"),
                      (3000, 128, 256, "c"), (2100, 200, 500, "c")]:
    if kind == "u":
        X = dg.dist_u(31 + n, n, d); Cn = dg.dist_u(33 + k, k, d)
    else:
        X = dg.dist_c(35 + n, n, d, max(4, k // 3), dg.default_sigma(d)); Cn = X[(np.arange(k) * 7919) % n].copy()
    if k >= 3:
        Cn[2] = Cn[0]
    a, md = capi.kmeans_assign(X, Cn, want_min_dist=True)
    want = co.assign_to_clusters(X, Cn)
    assert np.array_equal(a, want), (n, d, k)
    ref = np.array([co.squared_euclidean(X[i], Cn[int(want[i])]) for i in range(n)], dtype=np.float32)
    assert np.array_equal(bits(md), bits(ref)), (n, d, k)
pts, fb = capi.assign_stats(reset=True)
print("PRIM", pts, fb)

# (2) build_index end to end (iterations, convergence test, best-of-attempts) against the oracle
n, d, k = 6000, 96, 150
X = dg.dist_c(0x71, n, d, 450, dg.default_sigma(d))
init = mg.init_draws(7, 2, k, n)
ix = IVFFlatIndex.build_index(k, 2, 6, X, init_indices=init)
o = co.build_index(X, k, 2, 6, init)
assert np.array_equal(bits(ix.centroids), bits(o["centroids"]))
assert np.array_equal(ix.assignments, o["assignments"])
pts, fb = capi.assign_stats(reset=True)
print("BUILD", pts, fb)

# (3) NaN anywhere -> the reference's partial_cmp().unwrap() panic
Xn = dg.dist_u(5, 200, 32); Cn = dg.dist_u(6, 40, 32); Xn[150, 3] = np.nan
try:
    capi.kmeans_assign(Xn, Cn)
    raise SystemExit("NaN point not reported")
except capi.VersError as e:
    assert e.status == capi.ERR_NAN
Xn = dg.dist_u(5, 200, 32); Cn[17, 0] = np.nan
try:
    capi.kmeans_assign(Xn, Cn)
    raise SystemExit("NaN centroid not reported")
except capi.VersError as e:
    assert e.status == capi.ERR_NAN
# huge magnitudes: squares overflow to +inf -- nothing certifies, the exact scan decides
Xh = dg.dist_u(8, 300, 16) * np.float32(3e19); Ch = dg.dist_u(9, 10, 16) * np.float32(3e19)
assert np.array_equal(capi.kmeans_assign(Xh, Ch), co.assign_to_clusters(Xh, Ch))
print("DONE")
'''


def run(env_extra):
    env = dict(os.environ); env.update(env_extra); env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", BODY], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {l.split()[0]: tuple(int(v) for v in l.split()[1:]) for l in r.stdout.splitlines() if l[:4] in ("PRIM", "BUIL")}
    assert "DONE" in r.stdout
    return out


def test_matrix_core_assign_is_bit_exact():
    out = run({"VERS_OPTIONS": "assign=2"})
    pts, fb = out["PRIM"]
    assert pts == 5000 + 4097 + 777 + 300 + 129 + 9 + 3000 + 2100
    assert fb < pts // 2          # ties with a duplicate centroid and near-ties fail the certificate; most points pass
    pts, fb = out["BUILD"]
    assert pts >= 6000 * 2 * 2    # every assign pass of both attempts went through the matrix cores
    assert fb < pts // 10


def test_exact_scan_still_available():
    out = run({"VERS_OPTIONS": "assign=1"})
    assert out["PRIM"] == (0, 0) and out["BUILD"] == (0, 0)


def test_cascade_hi_only_first_filter_is_bit_exact():
    """option assign_terms = 1: ONE product of fp16 operands as the first filter of every pass (certificate as wide as the operands' measured
    residuals, the open points through the tile-limited exact re-scan), = 3: never; = 0 (default, the tests above): probed per build;
    assign_glds = 1 / 0: the LDS-DMA kernel on fp16 copies of both operands / the register-staged kernel, whatever k is.  Same bits always."""
    for opts in ("assign_terms=1,assign_glds=1", "assign_terms=3", "assign_terms=1,assign_glds=0"):
        out = run({"VERS_OPTIONS": f"assign=2,{opts}"})
        pts, _ = out["PRIM"]
        assert pts == 5000 + 4097 + 777 + 300 + 129 + 9 + 3000 + 2100
        pts, _ = out["BUILD"]
        assert pts >= 6000 * 2 * 2


GLDS_BODY = r'''
import numpy as np
from oracle import c_oracle as co
from tests import datagen as dg
from vers_amd import capi

def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

# shapes of dist_gemm_h_kernel (forced: assign_glds=1): fewer tiles than blocks, more tiles than blocks (a block walks several tiles and
# prefetches the next one's first K-tile under its epilogue), ragged last tiles in both directions, padding columns, both metrics' callers
for n, d, k, kind in [(700, 128, 256, "c"), (20000, 128, 1024, "c"), (33000, 256, 768, "u"), (5000, 200, 1500, "c"), (1300, 384, 2048, "c"), (9000, 640, 512, "c")]:
    if kind == "u":
        X = dg.dist_u(131 + n, n, d); Cn = dg.dist_u(133 + k, k, d)
    else:
        X = dg.dist_c(135 + n, n, d, max(4, k // 3), dg.default_sigma(d)); Cn = X[(np.arange(k) * 7919) % n].copy()
    Cn[2] = Cn[0]
    a, md = capi.kmeans_assign(X, Cn, want_min_dist=True)
    want = co.assign_to_clusters(X, Cn)
    assert np.array_equal(a, want), (n, d, k)
    idx = np.arange(0, n, max(1, n // 500))
    ref = np.array([co.squared_euclidean(X[i], Cn[int(want[i])]) for i in idx], dtype=np.float32)
    assert np.array_equal(bits(md[idx]), bits(ref)), (n, d, k)
pts, fb = capi.assign_stats(reset=True)
print("GLDS", pts, fb)
print("DONE")
'''


def test_lds_dma_contraction_shapes_are_bit_exact():
    """dist_gemm_h_kernel (the persistent fp16 x fp16 contraction staged by LDS-DMA) forced on at tile counts below, at and above the
    number of resident blocks: assignments and minimum distances == the oracle's."""
    env = dict(os.environ); env["VERS_OPTIONS"] = "assign=2,assign_terms=1,assign_glds=1"; env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", GLDS_BODY], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DONE" in r.stdout
    pts = int([l for l in r.stdout.splitlines() if l.startswith("GLDS")][0].split()[1])
    assert pts == 700 + 20000 + 33000 + 5000 + 1300 + 9000


def test_default_rules_at_4096_centroids_are_bit_exact():
    """No forcing beyond `assign=2` (matrix cores for a build this small): the cascade is probed, and from 4096 centroids on the LDS-DMA
    contraction is the default -- assignments and minimum distances == the oracle's."""
    body = GLDS_BODY.replace('[(700, 128, 256, "c"), (20000, 128, 1024, "c"), (33000, 256, 768, "u"), (5000, 200, 1500, "c"), (1300, 384, 2048, "c"), (9000, 640, 512, "c")]',
                             '[(12000, 128, 4096, "c"), (3000, 256, 4352, "u")]')
    assert "4096" in body
    env = dict(os.environ); env["VERS_OPTIONS"] = "assign=2"; env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DONE" in r.stdout
    pts = int([l for l in r.stdout.splitlines() if l.startswith("GLDS")][0].split()[1])
    assert pts == 12000 + 3000
