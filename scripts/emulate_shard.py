"""What EVERY rank of a W-GPU sharded search does per step, measured on one GPU (cfg3: N=10M d=768 nlist=4096 nprobe=32,
batches of 1024).  The index is built ONCE; for every world size W and every rank R of it a fresh handle takes the lists LPT
deals to (R, W) from the device-resident fields (vers_ivf_set_shard + vers_ivf_upload_dev) and search_partial_dev is timed on
the full 1024-query batches, one batch in flight and three (what bench.py runs).  A synchronous all-gather makes the step of
a W-GPU search the SLOWEST rank's: the table reports per-rank step, max and mean, and the rows each rank scanned.
The step is ONE library call per batch -- vers_ivf_search_sharded_dev: partial search -> the exchange on the batch's stream ->
merge of W partials -- and the exchange is, by default, a STAND-IN WITH RCCL's FOOTPRINT (EXCHANGE=standin,
vers_test_standin_gather): one kernel of WG workgroups (default 32) x 512 threads x 256 VGPRs + 37,664 B of LDS (SHAPE=256: torch's
librccl, 256 threads x 288 registers + 19,744 B; profiles/r05_rccl_kernel_meta.txt) that writes W partials and stays resident
for SPIN_US (default 25) -- what RCCL's device kernel needs of a CU while the peers' 1.1 MB travel.  Round 4 used a one-rank
ncclAllGather here (EXCHANGE=rccl1), which degenerates to a copy kernel and can sit beside a scan block; the real kernel cannot.
RESERVE="0,16,32": vers_set_option("scan_reserve_cus"): the persistent list scan leaves that many CUs to the exchange kernel and
to the other batches' coarse kernels.  EXCHANGE=0: the partial search alone, as in round 3.
usage: python scripts/emulate_shard.py [W ...]        (default 1 2 4 8; env STREAMS="1,3" STEPS=20 OUT=gpurun_out/emulate_shard.json)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import datagen as dg
from vers_amd import capi, testhooks
from vers_amd.index import IVFFlatIndex

worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
n = int(os.environ.get("ROWS", 10_000_000)); d = int(os.environ.get("DIM", 768)); nlist = int(os.environ.get("NLIST", 4096))
nprobe, B, top_k = 32, int(os.environ.get("BATCH", 1024)), 10
STREAMS = [int(s) for s in os.environ.get("STREAMS", "1,3").split(",")]
NSTEP = int(os.environ.get("STEPS", "20"))
EXCHANGE = os.environ.get("EXCHANGE", "standin")
if EXCHANGE == "1": EXCHANGE = "standin"
WG, SPIN_US, SHAPE = int(os.environ.get("WG", 32)), int(os.environ.get("SPIN_US", 25)), int(os.environ.get("SHAPE", 512))
RESERVES = [int(r) for r in os.environ.get("RESERVE", "-1").split(",")]   # -1 = the library's AUTO policy (production)
RANKS = os.environ.get("RANKS")  # e.g. "0,3": only these ranks of every world (quick looks)
SWEEP = os.environ.get("SWEEP")  # e.g. "seg_rows:0,320,448": the steps again with vers_set_option(name, value) for every value (keys tagged _<name><value>)
if SWEEP:
    SWEEP_NAME, SWEEP_VALS = SWEEP.split(":")[0], [int(v) for v in SWEEP.split(":")[1].split(",")]
    RESERVES = SWEEP_VALS
else:
    SWEEP_NAME = "scan_reserve_cus"
dev = torch.device("cuda:0")
X = torch.empty(n, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(X.data_ptr(), n, d, d, 1, 0x5EED0001, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
t0 = time.perf_counter()
ix0 = IVFFlatIndex(d, device=0)
ix0.build_dev(X.data_ptr(), n, nlist, 1, 4, init, want_fields=True)
t_build = time.perf_counter() - t0
lens = ix0.list_lengths()
Cd = torch.from_numpy(np.ascontiguousarray(ix0.centroids)).to(dev)
Ad = torch.from_numpy(ix0.assignments.astype(np.int64)).to(dev)
ix0.close(); del ix0
NQB = 8
Q = torch.empty(NQB * B, d, dtype=torch.float32, device=dev)
capi.gen_rows_dev(Q.data_ptr(), NQB * B, d, d, 1, 0x5EED0002, 0x5EEDC0DE, 16 * nlist, float(dg.default_sigma(d)))
max_s = max(STREAMS)
part_s = [torch.empty(2, B, top_k, dtype=torch.int64, device=dev) for _ in range(max_s)]   # [keys | ids] of this rank
keys_s = [p_[0] for p_ in part_s]; ids_s = [p_[1] for p_ in part_s]
res_s = [(torch.zeros(B, top_k, dtype=torch.int64, device=dev), torch.zeros(B, top_k, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(max_s)]
gat = None
import ctypes as C
from vers_amd import rccl
if EXCHANGE == "rccl1":
    gat = rccl.RcclComm(rccl.RcclComm.unique_id(), 0, 1, 0)


def standin(R, W):
    g = rccl.VersGather()
    testhooks.standin_gather(g, R, W, WG, SPIN_US, SHAPE, 37664 if SHAPE == 512 else 19744)
    return g
stream_objs = [torch.cuda.Stream() for _ in range(max_s)]
out = {"config": dict(rows=n, d=d, nlist=nlist, nprobe=nprobe, batch=B, top_k=top_k, steps=NSTEP, exchange=EXCHANGE,
                      standin=dict(workgroups=WG, spin_us=SPIN_US, threads=SHAPE, vgprs=256 if SHAPE == 512 else 288, lds_bytes=37664 if SHAPE == 512 else 19744) if EXCHANGE == "standin" else None,
                      scan_reserve_cus=RESERVES), "build_s": round(t_build, 2), "worlds": {}}
for W in worlds:
    owner = capi.shard_plan(lens, W) if W > 1 else np.zeros(nlist, np.uint8)
    ranks = range(W) if not RANKS else [int(r) for r in RANKS.split(",") if int(r) < W]
    rows_w = []
    for R in ranks:
        ix = IVFFlatIndex(d, device=0)
        if W > 1: ix.set_shard(R, W)
        ix.upload_dev(X.data_ptr(), n, d, Cd.data_ptr(), nlist, d, Ad.data_ptr())
        rec = {"rank": R, "stored_rows": int(lens[owner == R].sum())}
        sg = standin(R, W) if EXCHANGE == "standin" else None
        for RES in RESERVES:
          capi.set_option(SWEEP_NAME, RES)
          tag = "" if RES == RESERVES[0] else (f"_r{RES}" if not SWEEP else f"_{SWEEP_NAME}{RES}")
          for NS in STREAMS:
            streams = [torch.cuda.current_stream().cuda_stream] if NS == 1 else [so.cuda_stream for so in stream_objs[:NS]]
            allp_s = [torch.zeros(W, 2, B, top_k, dtype=torch.int64, device=dev) for _ in range(NS)] if EXCHANGE == "rccl1" else None
            def step(i):
                s = i % NS
                if EXCHANGE == "standin":  # ONE call: partial -> stand-in exchange kernel on the stream -> merge of W partials
                    # (W = 1: one GPU has no exchange -- g == NULL is vers_ivf_search_dev)
                    ix.search_sharded_dev(C.cast(C.byref(sg), capi._vp) if W > 1 else None, Q[(i % NQB) * B:].data_ptr(), d, B, top_k, nprobe,
                                          res_s[s][0].data_ptr(), res_s[s][1].data_ptr(), res_s[s][2].data_ptr(), streams[s])
                    return
                ix.search_partial_dev(Q[(i % NQB) * B:].data_ptr(), d, B, top_k, nprobe, keys_s[s].data_ptr(), ids_s[s].data_ptr(), streams[s])
                if EXCHANGE == "rccl1":  # the exchange's launch (one-rank RCCL all-gather of this rank's partial, on the stream) + the merge of W partials
                    g = gat._gather
                    assert g.all_gather_async(g.ctx, part_s[s].data_ptr(), allp_s[s].data_ptr(), 2 * B * top_k * 8, streams[s]) == 0
                    IVFFlatIndex.merge_partials_dev(allp_s[s].data_ptr(), allp_s[s].data_ptr() + 8 * B * top_k, 2 * B * top_k, W, B, top_k, nprobe,
                                                    res_s[s][0].data_ptr(), res_s[s][1].data_ptr(), res_s[s][2].data_ptr(), streams[s])
            for i in range(6): step(i)
            torch.cuda.synchronize(); _ = ix.scan_times(reset=True) if capi.env_option('scan_events', 2) != 0 else None; t0 = time.perf_counter()
            for i in range(NSTEP): step(6 + i)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / NSTEP
            rec[f"step_ms_s{NS}{tag}"] = round(dt * 1e3, 4)
            try:
                rec[f"scan_us_s{NS}{tag}"] = round(float(np.mean(ix.scan_times())) * 1e3, 1)
            except Exception:   # (option scan_events=0: no event records around the scans)
                rec[f"scan_us_s{NS}{tag}"] = float("nan")
            for s_ in streams: ix.poll(s_)
        capi.set_option(SWEEP_NAME, -1 if not SWEEP else RESERVES[0])
        # rows of the batch this rank scanned (the union of its probed lists), averaged over the NQB batches
        ur = []
        for i in range(NQB):
            ix.search_partial_dev(Q[i * B:].data_ptr(), d, B, top_k, nprobe, keys_s[0].data_ptr(), ids_s[0].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            try:
                ur.append(ix.last_scan()["union_rows"])
            except Exception:
                ur.append(0)
        rec["probed_rows"] = int(np.mean(ur))
        # the merge of the gathered partials (vers_topk_merge_dev): W copies of this rank's partial stand in for the all-gather's output
        allp = torch.empty(W, 2, B, top_k, dtype=torch.int64, device=dev)
        oi = torch.zeros(B, top_k, dtype=torch.int64, device=dev); od = torch.zeros(B, top_k, device=dev); oc = torch.zeros(B, dtype=torch.int32, device=dev)
        for r in range(W): allp[r, 0].copy_(keys_s[0]); allp[r, 1].copy_(ids_s[0])
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize(); tm0 = time.perf_counter()
        for i in range(50):
            IVFFlatIndex.merge_partials_dev(allp.data_ptr(), allp.data_ptr() + 8 * B * top_k, 2 * B * top_k, W, B, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
        torch.cuda.synchronize(); rec["merge_us"] = round((time.perf_counter() - tm0) / 50 * 1e6, 1)
        pst = ix.prescan_stats(); rec["rescanned_queries"] = int(pst["fallback_queries"])
        ix.close(); del ix
        rows_w.append(rec)
        print(json.dumps({"world": W, **rec}), flush=True)
    summ = {"ranks": rows_w}
    for RES in RESERVES:
      tag = "" if RES == RESERVES[0] else (f"_r{RES}" if not SWEEP else f"_{SWEEP_NAME}{RES}")
      for NS in STREAMS:
        v = np.array([r[f"step_ms_s{NS}{tag}"] for r in rows_w])
        summ[f"step_ms_s{NS}{tag}"] = {"max": float(v.max()), "mean": round(float(v.mean()), 4), "max_over_mean": round(float(v.max() / v.mean()), 3)}
        sc_ = np.array([r[f"scan_us_s{NS}{tag}"] for r in rows_w])
        summ[f"scan_us_s{NS}{tag}"] = {"max": float(np.nanmax(sc_)), "mean": round(float(np.nanmean(sc_)), 1)}
    pr = np.array([max(1, r["probed_rows"]) for r in rows_w], dtype=np.float64)
    summ["probed_rows"] = {"max": int(pr.max()), "mean": int(pr.mean()), "max_over_mean": round(float(pr.max() / pr.mean()), 3)}
    summ["all_gather_bytes_per_rank"] = 2 * B * top_k * 8
    out["worlds"][str(W)] = summ
    def tag_of(RES):
        return "" if RES == RESERVES[0] else (f"_r{RES}" if not SWEEP else f"_{SWEEP_NAME}{RES}")
    print(f"== world {W}: " + "  ".join(f"S={NS}{tag_of(RES)}: max {summ[f'step_ms_s{NS}' + tag_of(RES)]['max']:.3f} mean {summ[f'step_ms_s{NS}' + tag_of(RES)]['mean']:.3f} ms"
                                        for RES in RESERVES for NS in STREAMS) + f"  probed rows max/mean {summ['probed_rows']['max_over_mean']}", flush=True)
if "1" in out["worlds"]:
    for NS in STREAMS:
        one = out["worlds"]["1"][f"step_ms_s{NS}"]["max"]   # (one GPU: no exchange kernel to make room for, reserve 0)
        for RES in RESERVES:
            tag = "" if RES == RESERVES[0] else (f"_r{RES}" if not SWEEP else f"_{SWEEP_NAME}{RES}")
            out[f"predicted_speedup_s{NS}{tag}"] = {w: round(one / out["worlds"][w][f"step_ms_s{NS}{tag}"]["max"], 2) for w in out["worlds"]}
path = os.environ.get("EMU_OUT") or os.environ.get("OUT", "gpurun_out/emulate_shard.json")
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("predicted")}))
