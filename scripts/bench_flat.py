"""cfg2 micro-benchmark: brute-force scan N=1M d=128 f32, batch=1 (HBM-bound kernel).
Rotates several corpora so the 256 MiB Infinity Cache cannot serve the re-reads."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vers_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--b", type=int, default=1)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--corpora", type=int, default=4)
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(0)
ld = (args.d + 3) // 4 * 4
corp = []
for i in range(args.corpora):
    x = torch.randn(args.n, ld, device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    fc = capi.FlatCorpus(args.d)
    fc.upload_dev(x.data_ptr(), args.n, ld)
    fc._keep = x
    corp.append(fc)
q = torch.randn(args.b, args.d, device=dev); q = q / q.norm(dim=1, keepdim=True)
ids = torch.zeros(args.b, args.k, dtype=torch.int64, device=dev)
dist = torch.zeros(args.b, args.k, dtype=torch.float32, device=dev)
cnt = torch.zeros(args.b, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(fc):
    fc.search_dev(q.data_ptr(), args.d, args.b, args.k, 0, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), st)
for fc in corp: run(fc)
torch.cuda.synchronize()
scan_ms = []
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(args.iters):
    run(corp[it % len(corp)])
e1.record(); torch.cuda.synchronize()
tot = e0.elapsed_time(e1) / args.iters
for it in range(8):
    fc = corp[it % len(corp)]; run(fc); scan_ms.append(fc.last_scan_ms())
corp[0].poll(st)
scan = float(np.median(scan_ms))
on_shadow = args.b == 1 and args.k + 6 <= 64 and capi.env_option("shadow", 1) != 0 and capi.env_option("single_shadow", 1) != 0
bytes_ = args.n * (args.d * 2 + 4) if on_shadow else args.n * args.d * 4   # (the fp16 shadow row + its |x|^2, or the f32 row)
print("list scan on", "the fp16 shadow (flat1h_kernel)" if on_shadow else "the f32 rows (ordered chains)")
print(f"n={args.n} d={args.d} b={args.b}: call {tot*1e3:.1f} us/batch  scan kernel {scan*1e3:.1f} us  "
      f"-> scan {bytes_/scan/1e6:.1f} GB/s ({bytes_/scan/1e6/8000*100:.1f}% of 8 TB/s), end-to-end {args.b/tot*1e3:.0f} q/s")
# sanity vs torch
ref = ((corp[(args.iters+7) % len(corp)]._keep[:, :args.d] - q[0:1]) ** 2).sum(1).topk(args.k, largest=False)
print("ids match torch topk:", bool((ref.indices.sort().values == ids[0].sort().values).all()))
