"""What ONE rank of an N-GPU sharded search does per step, measured on one GPU: the full cfg3 index is built,
only the lists LPT gives to rank R of W are kept (vers_ivf_set_shard), and search_partial_dev is timed for the
full 1024-query batches.  No all-gather / merge (they need the other ranks): add ~0.1 ms for those.
usage: python scripts/emulate_shard.py W [R]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
W = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, d, nlist, nprobe, B, top_k = 10_000_000, 768, 4096, 32, 1024, 10
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0)
if W > 1: ix.set_shard(R, W)
ix.build_dev(X.data_ptr(), n, nlist, 1, 4, init)
del X
Q = torch.empty(8 * B, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), 8 * B, d, d, 1, 0x5EED0002, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
NS = int(os.environ.get("STREAMS", "1"))       # batches in flight: step i runs on stream i % NS with its own outputs
keys_s = [torch.empty(B, top_k, dtype=torch.int64, device=dev) for _ in range(NS)]
ids_s = [torch.empty(B, top_k, dtype=torch.int64, device=dev) for _ in range(NS)]
keys, ids = keys_s[0], ids_s[0]
_stream_objs = [torch.cuda.Stream() for _ in range(NS)] if NS > 1 else []
streams = [torch.cuda.current_stream().cuda_stream] if NS == 1 else [so.cuda_stream for so in _stream_objs]
st = streams[0]
AHEAD = os.environ.get("AHEAD", "0") != "0"   # next batch's coarse quantiser on the side stream (bench.py --ahead)
def step(i):
    s = i % NS
    if AHEAD: ix.coarse_ahead_dev(Q[((i + 1) % 8) * B:].data_ptr(), d, B, nprobe, streams[s])
    ix.search_partial_dev(Q[(i % 8) * B:].data_ptr(), d, B, top_k, nprobe, keys_s[s].data_ptr(), ids_s[s].data_ptr(), streams[s])
NSTEP = int(os.environ.get("STEPS", "20"))
for i in range(6): step(i)
torch.cuda.synchronize(); ix.scan_times(reset=True); t0 = time.perf_counter()
for i in range(NSTEP): step(6 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / NSTEP
# the merge of the gathered partials (vers_topk_merge_dev): W copies of this rank's partial stand in for the all-gather's
# output -- the kernel's time does not depend on whose keys they are
allp = torch.empty(W, 2, B, top_k, dtype=torch.int64, device=dev)
oi = torch.zeros(B, top_k, dtype=torch.int64, device=dev); od = torch.zeros(B, top_k, device=dev); oc = torch.zeros(B, dtype=torch.int32, device=dev)
for r in range(W): allp[r, 0].copy_(keys); allp[r, 1].copy_(ids)
torch.cuda.synchronize(); tm0 = time.perf_counter()
for i in range(50):
    IVFFlatIndex.merge_partials_dev(allp.data_ptr(), allp.data_ptr() + 8 * B * top_k, 2 * B * top_k, W, B, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
torch.cuda.synchronize(); t_merge = (time.perf_counter() - tm0) / 50
for s_ in streams: ix.poll(s_)
if os.environ.get("VERS_SCAN_DEBUG"): print("last scan:", ix.last_scan())   # (prints the phase stamps with VERS_SCAN_DEBUG=16)
pst = ix.prescan_stats()
print(f"world={W} rank={R} streams={NS}: {dt*1e3:.3f} ms per step for this rank (list scan {float(np.mean(ix.scan_times()))*1e3:.0f} us; "
      f"{pst['fallback_queries']} of {pst['batches'] * B} queries re-scanned exactly); merge of the {W} gathered partials {t_merge*1e6:.0f} us "
      f"(pipelined launches); all-gather payload {2 * B * top_k * 8} B per rank")
