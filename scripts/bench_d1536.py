"""The d = 1536 leg of bench.py (extra.d1536) on its own, for the profiler: N = 5M x 1536, nlist = 4096, nprobe = 32, batches of 1024
on the fp16 shadow -- prescan_kernel_g<true, 32, IvfSrc<32>, LO = false> (32-query blocks, query block as fp16 hi only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
n, d, nlist, nprobe, B, top_k = int(os.environ.get("ROWS", 5_000_000)), 1536, 4096, 32, 1024, 10
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001 + 0x1536, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
ix = IVFFlatIndex(d, device=0)
ix.build_dev(X.data_ptr(), n, nlist, 1, 4, (np.arange(nlist, dtype=np.uint64) * np.uint64(n // nlist)).astype(np.uint64))
del X; torch.cuda.empty_cache()
Q = torch.empty(4 * B, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), 4 * B, d, d, 1, 0x5EED0002 + 0x1536, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
oi = torch.zeros(B, top_k, dtype=torch.int64, device=dev); od = torch.zeros(B, top_k, device=dev); oc = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(5 + 20):
    if i == 5:
        torch.cuda.synchronize(); ix.scan_times(reset=True); t0 = time.perf_counter()
    ix.search_dev(Q[(i % 4) * B:].data_ptr(), d, B, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
ix.poll(st)
ls = ix.last_scan(); ms = float(np.mean(ix.scan_times()))
by = ls["union_rows"] * (2 * d + 4) + nlist * d * 4
print(f"d=1536 N={n}: step {dt * 1e3:.3f} ms, list scan {ms:.3f} ms, algorithmic {by / 1e9:.2f} GB = {by / (ms * 1e-3) / 8e12:.3f} of 8 TB/s, streamed / union rows "
      f"{ls['streamed_rows'] / max(1, ls['union_rows']):.3f}, re-scanned queries {ix.prescan_stats()['fallback_queries']}")
