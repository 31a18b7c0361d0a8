"""The edges of the fast domain on the headline's index (cfg3: N=10M d=768 nlist=4096), one batch in flight: nprobe beyond a key per
lane, wide results, and the same steps under the compact memory layout.  usage: python scripts/bench_edges.py [ROWS=10000000] [MEMORY=0|1] [SHAPES=name:batch:top_k:nprobe,...] [AB=option]
Prints one JSON object per shape on stdout."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi
from vers_amd.index import IVFFlatIndex

kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n, d, nlist = int(kv.get("ROWS", 10_000_000)), int(kv.get("D", 768)), int(kv.get("NLIST", 4096))
if "MEMORY" in kv:
    capi.set_option("memory", int(kv["MEMORY"]))
dev = torch.device("cuda:0")
SEED_X, SEED_Q, SEED_C = 0x5EED0001, 0x5EED0002, 0x5EEDC0DE
n_modes, sigma = 16 * nlist, float(dg.default_sigma(d))
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, SEED_X, SEED_C, n_modes, sigma)
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
ix = IVFFlatIndex(d, device=0)
capi.mem_stats(reset_peak=True)
t0 = time.perf_counter(); ix.build_dev(X.data_ptr(), n, nlist, 1, 4, init); tb = time.perf_counter() - t0
del X; torch.cuda.empty_cache()
now, peak = capi.mem_stats()
print(json.dumps({"build_s": round(tb, 2), "library_bytes": int(now), "bytes_per_row": round(now / n, 1), "over_f32_rows": round(now / n / (4 * d), 3),
                  "layout": ix.layout_bytes(), "peak_over_rows": round(peak / (n * d * 4), 3)}), flush=True)
Q = torch.empty(8 * 1024, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), 8 * 1024, d, d, 1, SEED_Q, SEED_C, n_modes, sigma)
st = torch.cuda.current_stream().cuda_stream
shapes = [("headline", 1024, 10, 32), ("nprobe_64", 256, 10, 64), ("nprobe_65", 256, 10, 65), ("nprobe_128", 256, 10, 128), ("nprobe_256", 256, 10, 256),
          ("nprobe_128_b1024", 1024, 10, 128), ("top_k_20", 256, 20, 32), ("top_k_30", 256, 30, 32), ("top_k_36", 256, 36, 32), ("top_k_40", 256, 40, 32), ("top_k_48", 256, 48, 32),
          ("top_k_58", 256, 58, 32), ("top_k_64", 256, 64, 32), ("top_k_100", 256, 100, 32), ("top_k_128", 256, 128, 32), ("b1", 1, 10, 32)]
if "SHAPES" in kv:   # name:batch:top_k:nprobe,...
    shapes = [(a, int(b_), int(c), int(e)) for a, b_, c, e in (x.split(":") for x in kv["SHAPES"].split(","))]
only = kv.get("ONLY")
ab = kv.get("AB")    # AB=option: every shape with the option at 0, then at 1
for name, b, tk, npb, opt in [(sh + (v,)) for sh in shapes for v in ((0, 1) if ab else (None,))]:
    if only and name not in only.split(","):
        continue
    if ab:
        capi.set_option(ab, opt)
    oi = torch.zeros(b, tk, dtype=torch.int64, device=dev); od = torch.zeros(b, tk, device=dev); oc = torch.zeros(b, dtype=torch.int32, device=dev)
    def step(i):
        ix.search_dev(Q[(i * b) % (8 * 1024 - b + 1):].data_ptr(), d, b, tk, npb, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
    for i in range(3): step(i)
    torch.cuda.synchronize(); pb0 = ix.prescan_stats()["batches"]; fb0 = ix.prescan_stats()["fallback_queries"]; nst = 8 if b > 1 else 200; ix.scan_times(reset=True); t0 = time.perf_counter()
    for i in range(nst): step(3 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nst
    ix.poll(st)
    rec = {"shape": name, **({ab: opt} if ab else {}), "batch": b, "top_k": tk, "nprobe": npb, "us_per_batch": round(dt * 1e6, 1), "queries_per_sec": round(b / dt, 1),
           "matrix_core_batches": ix.prescan_stats()["batches"] - pb0, "of": nst,
           "rescanned_queries_per_batch": round((ix.prescan_stats()["fallback_queries"] - fb0) / nst, 1)}
    for key, fn in (("list_scan_us", lambda: round(float(np.mean(ix.scan_times(reset=True))) * 1e3, 1)), ("union_rows", lambda: int(ix.last_scan()["union_rows"])),
                    ("coarse_ms", lambda: ix.last_coarse_ms()), ("finish_us", lambda: round(ix.last_finish_ms() * 1e3, 1))):
        try:
            rec[key] = fn()
        except Exception:
            pass
    print(json.dumps(rec), flush=True)
