import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex
n, d, nlist, B, nprobe = int(os.environ.get("ROWS", 600000)), 768, int(os.environ.get("NLIST", 256)), 512, 16
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0)
ix.build_dev(X.data_ptr(), n, nlist, 1, 2, init)
Q = torch.empty(B, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), B, d, d, 1, 0x5EED0002, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
ids = torch.zeros(B, 10, dtype=torch.int64, device=dev); dst = torch.zeros(B, 10, dtype=torch.float32, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ix.search_dev(Q.data_ptr(), d, B, 10, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st); ix.poll(st)
np.savez(sys.argv[1], ids=ids.cpu().numpy(), dst=dst.cpu().numpy().view(np.uint32))
print(ix.shadow_state(), ix.prescan_stats())
