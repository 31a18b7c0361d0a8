"""Latency of the drop-in call itself: vers_ivf_search with HOST pointers, one query per call (what
Index::search_approximate does in vers), reference semantics (nprobe = 0) and nprobe = 32, at cfg3 geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.capi import lib, check, _ptr
from vers_amd.index import IVFFlatIndex

n, d, nlist = int(os.environ.get("ROWS", 10_000_000)), 768, 4096
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0)
ix.build_dev(X.data_ptr(), n, nlist, 1, 2, init)
del X
Q = dg.dist_c(0x77, 256, d, 16 * nlist, dg.default_sigma(d)).astype(np.float32)
ids = np.zeros(10, dtype=np.uint64); dist = np.zeros(10, dtype=np.float32); cnt = np.zeros(1, dtype=np.uint32)
for nprobe, spin in ((32, 1), (0, 1), (32, 1), (32, 0), (32, 1), (32, 0), (0, 1), (0, 0)):  # (the first loop after the build runs ~2x slower whatever the mode: discard it)
    capi.set_option("host_spin", spin)   # 1 (default): spin on the pinned status word; 0: hipStreamSynchronize (rounds 1-4)
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(256):
            check(lib().vers_ivf_search(ix._h, _ptr(Q[i]), 4 * d, 1, 10, nprobe, _ptr(ids), _ptr(dist), _ptr(cnt)))
        dt = (time.perf_counter() - t0) / 256
    print(f"host-pointer single-query call, nprobe={nprobe}, host_spin={spin}: {dt*1e6:.1f} us per call ({1/dt:.0f} q/s)")
    if spin == 0: continue
    # the same through device pointers, synchronised per call, and pipelined (no sync between calls)
    qd = torch.from_numpy(Q).to(dev); idd = torch.zeros(10, dtype=torch.int64, device=dev)
    dd = torch.zeros(10, dtype=torch.float32, device=dev); cd = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for sync_each in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(256):
            ix.search_dev(qd[i].data_ptr(), d, 1, 10, nprobe, idd.data_ptr(), dd.data_ptr(), cd.data_ptr(), st)
            if sync_each: torch.cuda.synchronize()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 256
        print(f"   device pointers, {'sync per call' if sync_each else 'pipelined'}: {dt*1e6:.1f} us per call")
