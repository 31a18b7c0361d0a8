"""The k-means assign contraction as build_index runs it (vers_build_stats: HIP events around the launches), at cfg3's and
cfg5's cluster counts.  usage: [VERS_OPTIONS=assign_terms=3] [ONLY=0] python scripts/bench_assign.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
d = 768
dev = torch.device("cuda:0")
CASES = ((4_194_304, 4096, 2), (1_048_576, 65536, 1))
if os.environ.get("ONLY"): CASES = (CASES[int(os.environ["ONLY"])],)   # ONLY=0: cfg3's cluster count alone (quick PMC looks)
for n, k, iters in CASES:
    X = torch.empty(n, d, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * 4096, float(dg.default_sigma(d)))
    init = (dg.mix64(np.uint64(0xB01D) + np.arange(k, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
    ix = IVFFlatIndex(d, device=0)
    capi.build_stats(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.build_dev(X.data_ptr(), n, k, 1, iters, init)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bs = capi.build_stats(reset=True)
    tf = bs["gemm_flop"] / (bs["gemm_ms"] * 1e-3) / 1e12
    print(f"N={n} k={k}: build {dt:.3f} s; contraction {bs['gemm_ms'] / bs['gemm_launches'] * 1e3:.1f} us per launch x {int(bs['gemm_launches'])} = {tf:.1f} algorithmic TFLOP/s; "
          f"assign pass {bs['assign_ms'] / bs['assign_passes']:.1f} ms (whole, incl. split / re-score / re-scan); redone {100 * bs['redone_points'] * 2 * k * d / bs['gemm_flop']:.2f} %; "
          f"cost bits {np.float32(ix.cost).view(np.uint32):#010x}", flush=True)
    ix.close(); del X; torch.cuda.empty_cache()
