"""cfg3 (N=10M d=768 nlist=4096 nprobe=32, top_k=10) at batch 1 .. 1024: queries/s, us per batch and which list scan ran, with the
small-batch switch at its default and forced either way -- the data behind vers_set_option("pre_min_batch").  Same-process A/B.
usage: python scripts/batch_sweep.py      (env ROWS DIM NLIST OUT)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
n = int(os.environ.get("ROWS", 10_000_000)); d = int(os.environ.get("DIM", 768)); nlist = int(os.environ.get("NLIST", 4096)); nprobe, top_k = 32, 10
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0); ix.build_dev(X.data_ptr(), n, nlist, 1, 4, init); del X
Q = torch.empty(8192, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), 8192, d, d, 1, 0x5EED0002, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
S = 3
streams = [torch.cuda.Stream() for _ in range(S)]
outs = [(torch.zeros(1024, top_k, dtype=torch.int64, device=dev), torch.zeros(1024, top_k, device=dev), torch.zeros(1024, dtype=torch.int32, device=dev)) for _ in range(S)]
def run(b, steps):
    def step(i):
        o = outs[i % S]
        ix.search_dev(Q[(i * b) % (8192 - b):].data_ptr(), d, b, top_k, nprobe, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), streams[i % S].cuda_stream)
    for i in range(5): step(i)
    torch.cuda.synchronize(); b0 = ix.prescan_stats()["batches"]; t0 = time.perf_counter()
    for i in range(steps): step(5 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    for s in streams: ix.poll(s.cuda_stream)
    return dt, ix.prescan_stats()["batches"] - b0 == steps
res = {}
for mode, val in (("matrix_cores_from_batch_2", 2), ("ordered_chains_below_2_queries_per_list", 1 << 30), ("default", 4)):
    capi.set_option("pre_min_batch", val)
    res[mode] = {}
    for b in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
        dt, pre = run(b, 60 if b <= 128 else 30)
        res[mode][str(b)] = {"us_per_batch": round(dt * 1e6, 1), "queries_per_sec": round(b / dt, 1), "list_scan": "matrix cores (fp16 shadow) + exact finish" if pre else ("single-query records" if b == 1 else "ordered chains")}
        print(mode, b, res[mode][str(b)], flush=True)
path = os.environ.get("OUT", "gpurun_out/batch_sweep.json"); os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump({"config": dict(rows=n, d=d, nlist=nlist, nprobe=nprobe, top_k=top_k, batches_in_flight=S), "sweep": res}, open(path, "w"), indent=1)
