"""build_index at cfg5's cluster count on one GPU (scaled-down N): where does a k-means pass go when k = 65536."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
n, d, k = int(os.environ.get("ROWS", 2_000_000)), 768, int(os.environ.get("K", 65536))
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 4 * k, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(k, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
ix.build_dev(X.data_ptr(), n, k, 1, 1, init)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pts, fb = capi.assign_stats()
print(f"N={n} k={k} d={d}: build (1 iteration + final assign = 2 assign passes) {dt:.2f} s -> {2*2*n*k*d/dt/1e12:.1f} TFLOP/s over the whole build; matrix-core assign {pts} points, {fb} re-done exactly; cost {float(ix.cost):.1f}")
