"""cfg4 / cfg5 at ONE RANK'S NOMINAL SIZE on one GPU (no 8-GPU node has ever been available: SCALE skipped in every round).

cfg4 (BASELINE.json configs[3]: IVFFlat N=100M d=768 sharded over 8 GPUs): what rank R of 8 holds and does.
  The corpus never exists anywhere: it is generated in chunks by the counter-based generator, each chunk is assigned to a
  quantiser trained on the first `sample` rows (vers_kmeans_assign_dev: assign_to_clusters, ivfflat.rs:29-46) and STREAMED into a
  handle sharded as rank R of `world` (vers_ivf_upload_begin / _chunk_dev / _end: the reference's load_index -> search
  sequence, base.rs:45-58 / utils.rs:140-148, with no buffer of n_total rows).  Then the rank's step of the sharded search
  (partial search + exchange launch + merge of `world` partials) is timed with 1 and 3 batches in flight, and the rank's
  partial results are compared BITWISE with the CPU restatement (oracle/vers_oracle.c: vo_search_nprobe) run over the
  rank's own sub-index (all centroids, the lists this rank owns among the probed ones) on `check` sampled queries.
cfg5 (configs[4]: k-means build_index N=50M d=768 k=65536 on 8 GPUs, row-sharded): one rank's 6.25M x 768 rows through
  vers_ivf_build_sharded_dev (world 1: the rank's own work of every pass -- assign, grouping, running sums, cost fold, lists;
  the chain hops need the peers), max_iterations passes + the final assign.

usage: python scripts/rank_nominal.py cfg4|cfg5 [key=value ...]   (ROWS NLIST RANK WORLD CHUNK SAMPLE CHECK STEPS ITERS K OUT)
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SEED_X, SEED_Q, SEED_C = 0x5EED0001, 0x5EED0002, 0x5EEDC0DE


def _gb(x):
    return round(x / 1e9, 2)


def cfg4_rank(dev_index=0, rows=100_000_000, d=768, nlist=16384, rank=0, world=8, chunk=2_000_000, sample=10_000_000, iters=2,
              nprobe=32, B=1024, top_k=10, steps=20, check=32, streams=(1, 3), modes_per_list=16, log=print, exchange=True):
    import torch
    from tests import datagen as dg
    from vers_amd import capi, testhooks
    from vers_amd.index import IVFFlatIndex
    dev = torch.device(f"cuda:{dev_index}")
    ld = (d + 3) // 4 * 4
    n_modes = max(1, modes_per_list * nlist)
    sigma = float(dg.default_sigma(d))
    sample = min(sample, rows)
    out = {"workload": f"rank {rank} of {world} of IVFFlat N={rows} d={d} nlist={nlist} nprobe={nprobe} batch={B} top_k={top_k} (cfg4 at one rank's nominal size, one GPU)",
           "rows_total": rows, "nlist": nlist, "rank": rank, "world": world, "chunk_rows": chunk}
    capi.mem_stats(reset_peak=True)
    # ---- 1. the quantiser: k-means on the first `sample` rows (the reference would train on all rows: ivfflat.rs:102-121; a
    #         100M-row k-means is cfg5's subject, not this leg's) --------------------------------------------------------------------
    t0 = time.perf_counter()
    Xs = torch.empty(sample, ld, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(Xs.data_ptr(), sample, d, ld, 1, SEED_X, SEED_C, n_modes, sigma, start_row=0)
    init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(sample)).astype(np.uint64)
    ix0 = IVFFlatIndex(d, device=dev_index)
    ix0.build_dev(Xs.data_ptr(), sample, nlist, 1, iters, init, want_fields=True)
    cent = np.ascontiguousarray(ix0.centroids)
    ix0.close(); del ix0, Xs
    torch.cuda.empty_cache()
    out["train_s"] = round(time.perf_counter() - t0, 2)
    out["train"] = f"k-means on rows [0, {sample}) ({iters} iterations, injected init draws), then every chunk assigned to it"
    log(f"[cfg4_rank] quantiser trained on {sample} rows in {out['train_s']} s")
    # ---- 2. pass 1 over the corpus: assignments + list lengths (never more than one chunk of rows resident) ----------------------
    Cd = torch.from_numpy(cent).to(dev)
    A = torch.empty(rows, dtype=torch.int64, device=dev)      # the reference's `assignments` (usize), 8 B per row
    Xc = torch.empty(min(chunk, rows), ld, dtype=torch.float32, device=dev)
    t0 = time.perf_counter(); t_gen = 0.0
    for a in range(0, rows, chunk):
        m = min(chunk, rows - a)
        tg = time.perf_counter()
        capi.gen_rows_dev(Xc.data_ptr(), m, d, ld, 1, SEED_X, SEED_C, n_modes, sigma, start_row=a)
        torch.cuda.synchronize(); t_gen += time.perf_counter() - tg
        capi.kmeans_assign_dev(Xc.data_ptr(), m, ld, Cd.data_ptr(), nlist, d, d, A[a:].data_ptr(), device=dev_index)
    t_assign = time.perf_counter() - t0 - t_gen
    capi.kmeans_assign_release(dev_index)   # (the quantiser's device scratch: not part of the index's footprint below)
    lens = torch.bincount(A, minlength=nlist).cpu().numpy().astype(np.uint64)
    out["assign_pass"] = {"seconds": round(t_assign, 2), "algorithmic_tflops": round(2.0 * rows * nlist * d / t_assign / 1e12, 1),
                          "generator_seconds": round(t_gen, 2)}
    out["list_len"] = {"min": int(lens.min()), "mean": round(float(lens.mean()), 1), "max": int(lens.max()), "empty": int((lens == 0).sum())}
    log(f"[cfg4_rank] {rows} rows assigned to {nlist} lists in {t_assign:.1f} s ({out['assign_pass']['algorithmic_tflops']} TFLOP/s), lists {out['list_len']}")
    # ---- 3. pass 2: the streamed upload into rank `rank` of `world` -----------------------------------------------------------------
    ix = IVFFlatIndex(d, device=dev_index)
    if world > 1:
        ix.set_shard(rank, world)
    mem0, _ = capi.mem_stats(reset_peak=True)
    t0 = time.perf_counter(); t_gen2 = 0.0
    ix.upload_begin(cent, lens, rows)
    t_begin = time.perf_counter() - t0
    for a in range(0, rows, chunk):
        m = min(chunk, rows - a)
        tg = time.perf_counter()
        capi.gen_rows_dev(Xc.data_ptr(), m, d, ld, 1, SEED_X, SEED_C, n_modes, sigma, start_row=a)
        torch.cuda.synchronize(); t_gen2 += time.perf_counter() - tg
        ix.upload_chunk_dev(Xc.data_ptr(), ld, A[a:].data_ptr(), a, m)
    t_chunks = time.perf_counter() - t0 - t_begin - t_gen2
    te = time.perf_counter()
    ix.upload_end()
    t_end = time.perf_counter() - te
    t_up = time.perf_counter() - t0 - t_gen2
    mem_now, mem_peak = capi.mem_stats()
    owner = ix.owners()
    stored = int(lens[owner == rank].sum()) if world > 1 else int(lens.sum())
    lay = ix.layout_bytes()
    sh = ix.shadow_state() if hasattr(ix, "shadow_state") else None
    out["upload"] = {"seconds": round(t_up, 2), "begin_s": round(t_begin, 3), "chunks_s": round(t_chunks, 3), "end_s": round(t_end, 3), "stored_rows": stored, "stored_lists": int((owner == rank).sum()) if world > 1 else nlist,
                     "library_bytes_now": int(mem_now - mem0), "library_bytes_peak": int(mem_peak - mem0),
                     "f32_tile_rows_bytes": int(lay["rows"]), "fp16_shadow_bytes": int(lay["shadow"]), "rowmajor_bytes": int(lay["rowmajor"]),
                     "rowmajor_kept": bool(lay["rowmajor"] > 0), "shadow_kept": bool(lay["shadow"] > 0),
                     "peak_over_stored_f32_bytes": round((mem_peak - mem0) / max(1, stored * d * 4), 3),
                     "note": "vers_ivf_upload_begin / _chunk_dev / _end: the rank keeps only its lists' rows; no buffer of n_total rows "
                             "(device memory besides the index: one chunk of rows + 8 B per row of assignments, both the caller's)"}
    log(f"[cfg4_rank] streamed upload: {stored} rows kept of {rows} in {t_up:.1f} s; library holds {_gb(mem_now - mem0)} GB (peak {_gb(mem_peak - mem0)} GB), "
        f"row-major copy kept: {lay['rowmajor'] > 0}, shadow kept: {lay['shadow'] > 0}")
    del Xc
    torch.cuda.empty_cache()
    # ---- 4. the rank's step of the sharded search -----------------------------------------------------------------------------------
    NQB = 4
    Q = torch.empty(NQB * B, ld, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(Q.data_ptr(), NQB * B, d, ld, 1, SEED_Q, SEED_C, n_modes, sigma)
    max_s = max(streams)
    part_s = [torch.empty(2, B, top_k, dtype=torch.int64, device=dev) for _ in range(max_s)]
    res_s = [(torch.zeros(B, top_k, dtype=torch.int64, device=dev), torch.zeros(B, top_k, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(max_s)]
    # The step is ONE library call per batch (vers_ivf_search_sharded_dev: partial search -> exchange on the batch's stream -> merge of
    # `world` partials) and the exchange is the STAND-IN WITH RCCL's FOOTPRINT (vers_test_standin_gather: 32 workgroups x 512 threads x
    # 256 VGPRs + 37,664 B of LDS, resident for 25 us, writing `world` partials) -- what the real all-gather needs of the chip while the
    # peers' bytes travel; the peers themselves need the other GPUs.
    sg = None
    if exchange and world > 1:
        from vers_amd import rccl as vrccl   # (the struct definition only: no RCCL call is made)
        sg = vrccl.VersGather()
        testhooks.standin_gather(sg, rank, world, 32, 25, 512, 37664)
    gp = C.cast(C.byref(sg), capi._vp) if sg is not None else None
    stream_objs = [torch.cuda.Stream(device=dev) for _ in range(max_s)]
    step_ms, scan_us = {}, {}
    for NS in streams:
        sts = [torch.cuda.current_stream(dev).cuda_stream] if NS == 1 else [so.cuda_stream for so in stream_objs[:NS]]

        def step(i):
            s = i % NS
            ix.search_sharded_dev(gp, Q[(i % NQB) * B:].data_ptr(), ld, B, top_k, nprobe, res_s[s][0].data_ptr(), res_s[s][1].data_ptr(), res_s[s][2].data_ptr(), sts[s])
        for i in range(5):
            step(i)
        torch.cuda.synchronize(); ix.scan_times(reset=True); t0 = time.perf_counter()
        for i in range(steps):
            step(5 + i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        step_ms[f"s{NS}"] = round(dt * 1e3, 4)
        scan_us[f"s{NS}"] = round(float(np.mean(ix.scan_times())) * 1e3, 1)
        for s_ in sts:
            ix.poll(s_)
    # the bytes the scan is measured on: the union of this rank's probed lists (counted on the device), one batch in flight
    ur, sc = [], []
    st0 = torch.cuda.current_stream(dev).cuda_stream
    for i in range(NQB):
        ix.search_partial_dev(Q[i * B:].data_ptr(), ld, B, top_k, nprobe, part_s[0][0].data_ptr(), part_s[0][1].data_ptr(), st0)
        torch.cuda.synchronize()
        ls = ix.last_scan(); ur.append(ls["union_rows"]); sc.append(ls["ms"])
    shadow = lay["shadow"] > 0
    per_row = (2 * d + 4) if shadow else 4 * d
    by = float(np.mean(ur)) * per_row + nlist * d * 4
    scan_ms = float(np.mean(sc))
    pst = ix.prescan_stats()
    out["search"] = {"step_ms": step_ms, "list_scan_us_in_step": scan_us, "queries_per_sec_if_every_rank_like_this": {k_: round(B / (v / 1e3), 1) for k_, v in step_ms.items()},
                     "list_scan_alone_ms": round(scan_ms, 4), "probed_rows_union": int(np.mean(ur)), "algorithmic_bytes": int(by),
                     "list_scan_frac_of_8TBs": round(by / (scan_ms / 1e3) / 8e12, 4), "row_operand": "fp16 shadow" if shadow else "f32 rows",
                     "rescanned_queries": int(pst["fallback_queries"]), "matrix_core_batches": int(pst["batches"]),
                     "step_includes": "vers_ivf_search_sharded_dev: partial search + " + ("exchange stand-in with RCCL's footprint (32 x 512 threads x 256 VGPRs, 37,664 B LDS, 25 us) + " if sg is not None else "") + f"merge of {world} partials"}
    log(f"[cfg4_rank] step {step_ms} ms; list scan alone {scan_ms:.3f} ms = {out['search']['list_scan_frac_of_8TBs']} of peak on {_gb(by)} GB; re-scanned queries {pst['fallback_queries']}")
    # ---- 5. GPU == CPU restatement, bit for bit, over the rank's own sub-index ---------------------------------------------------------
    if check:
        from oracle import c_oracle as co
        fp_, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64)
        ix.search_partial_dev(Q.data_ptr(), ld, B, top_k, nprobe, part_s[0][0].data_ptr(), part_s[0][1].data_ptr(), st0)
        oi, od, oc = res_s[0]
        IVFFlatIndex.merge_partials_dev(part_s[0][0].data_ptr(), part_s[0][1].data_ptr(), B * top_k, 1, B, top_k, nprobe, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st0)
        torch.cuda.synchronize(); ix.poll(st0)
        ids_h = oi.cpu().numpy().astype(np.uint64); dst_h = od.cpu().numpy(); cnt_h = oc.cpu().numpy()
        qh = Q[:B, :d].cpu().numpy()
        pick = np.unique(np.linspace(0, B - 1, check).astype(np.int64))
        bad, t_cpu, rows_cpu = 0, 0.0, 0
        for qi in pick:
            q = np.ascontiguousarray(qh[qi])
            ranked, _ = co.search_exhaustive(cent, q, nlist)
            lists = []
            for c in ranked[:nprobe]:
                if world > 1 and owner[int(c)] != rank:
                    continue
                r_, id_ = ix.get_list(int(c)); lists.append((int(c), r_, id_))
            off = np.zeros(nlist + 1, dtype=np.uint64)
            for c, r_, id_ in lists:
                off[c + 1] = len(id_)
            off = np.cumsum(off).astype(np.uint64)
            tot = max(1, int(off[-1]))
            vals = np.zeros((tot, d), dtype=np.float32); vid = np.zeros(tot, dtype=np.uint64)
            for c, r_, id_ in lists:
                vals[int(off[c]):int(off[c + 1])] = r_; vid[int(off[c]):int(off[c + 1])] = id_
            loc = np.arange(tot, dtype=np.uint64)
            gi = np.zeros(top_k, dtype=np.uint64); gd = np.zeros(top_k, dtype=np.float32)
            t0 = time.perf_counter()
            m = co.lib().vo_search_nprobe_m(vals.ctypes.data_as(fp_), cent.ctypes.data_as(fp_), nlist, d, off.ctypes.data_as(u64p), loc.ctypes.data_as(u64p),
                                            q.ctypes.data_as(fp_), top_k, nprobe, gi.ctypes.data_as(u64p), gd.ctypes.data_as(fp_), 0)
            t_cpu += time.perf_counter() - t0; rows_cpu += int(off[-1])
            ok = m >= 0 and cnt_h[qi] == m and np.array_equal(vid[gi[:m].astype(np.int64)], ids_h[qi, :m]) and np.array_equal(gd[:m].view(np.uint32), dst_h[qi, :m].view(np.uint32))
            bad += 0 if ok else 1
        out["check"] = {"queries": int(len(pick)), "gpu_matches_cpu_bitwise": bad == 0, "mismatching_queries": bad,
                        "cpu_queries_per_sec_1_core": round(len(pick) / max(t_cpu, 1e-9), 2), "rows_scored_on_cpu": rows_cpu,
                        "how": "vo_search_nprobe (oracle/vers_oracle.c) per query over the rank's sub-index: all centroids, the lists this rank owns among "
                               "the query's nprobe nearest, read back from HBM; compared with the rank's partial top-k (ids, count, distance bits)"}
        log(f"[cfg4_rank] GPU == CPU bitwise on {len(pick)} queries over the rank's sub-index: {bad == 0}")
    ix.close()
    del A, Q
    torch.cuda.empty_cache()
    return out


class _OneRankComm(C.Structure):
    """vers_comm_t of a world of one (include/vers_hip.h): no callback is ever called."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather", C.c_void_p), ("send", C.c_void_p),
                ("recv", C.c_void_p), ("broadcast", C.c_void_p), ("all_to_all_v", C.c_void_p)]

    def ptr(self):
        return C.byref(self)


def cfg5_rank(dev_index=0, rows_total=50_000_000, world=8, d=768, k=65536, iters=1, modes_per_list=4, log=print):
    import torch
    from tests import datagen as dg
    from vers_amd import capi
    from vers_amd.index import IVFFlatIndex
    dev = torch.device(f"cuda:{dev_index}")
    n = rows_total // world
    ld = (d + 3) // 4 * 4
    n_modes = max(1, modes_per_list * k)
    sigma = float(dg.default_sigma(d))
    out = {"workload": f"one rank's share of k-means build_index N={rows_total} d={d} k={k} row-sharded over {world}: {n} rows through vers_ivf_build_sharded_dev "
                       f"(world 1), {iters} iteration(s) + the final assign (cfg5 at one rank's nominal size, one GPU)", "rows": n, "k": k}
    X = torch.empty(n, ld, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(X.data_ptr(), n, d, ld, 1, SEED_X, SEED_C, n_modes, sigma, start_row=0)
    init = (dg.mix64(np.uint64(0xB01D) + np.arange(k, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
    torch.cuda.synchronize()
    ix = IVFFlatIndex(d, device=dev_index)
    mem0, _ = capi.mem_stats(reset_peak=True)
    capi.build_stats(reset=True); capi.build_phases(reset=True); capi.assign_stats(reset=True)
    comm = _OneRankComm(None, 0, 1, None, None, None, None, None)
    t0 = time.perf_counter()
    kept = ix.build_sharded_dev(X.data_ptr(), n, ld, 0, n, k, 1, iters, init, comm, want_assignments=True)
    t_build = time.perf_counter() - t0
    mem_now, mem_peak = capi.mem_stats()
    bs, bph = capi.build_stats(), capi.build_phases()
    ast = capi.assign_stats()
    passes = max(1.0, bs["assign_passes"])
    lens = ix.list_lengths()
    lay = ix.layout_bytes()
    out.update({"kept": bool(kept), "iterations": int(ix.iterations[0]), "build_s": round(t_build, 2), "cost": float(ix.cost),
                "assign_passes": int(passes), "seconds_per_assign_pass": round(bs["assign_ms"] / passes / 1e3, 3),
                "contraction_algorithmic_tflops": round(bs["gemm_flop"] / max(bs["gemm_ms"], 1e-9) / 1e9, 1),
                "assign_pass_algorithmic_tflops": round(2.0 * n * k * d * passes / max(bs["assign_ms"], 1e-9) / 1e9, 1),
                "redone_points_frac": round(bs["redone_points"] / max(1.0, n * passes), 5),
                "phases_ms": bph, "update_ms_per_pass": round(bs["update_ms"] / max(1, int(ix.iterations[0])), 1), "cost_fold_ms": round(bs["cost_ms"], 1),
                "library_bytes_peak": int(mem_peak - mem0), "library_bytes_now": int(mem_now - mem0), "rows_bytes": int(n) * d * 4,
                "peak_over_rows_bytes": round((mem_peak - mem0) / (n * d * 4.0), 3),
                "rowmajor_kept": bool(lay["rowmajor"] > 0), "shadow_kept": bool(lay["shadow"] > 0),
                "chain_hop_bytes": int(k) * ld * 4, "chain_hops_per_pass_at_world_8": world - 1,
                "chain_note": f"on {world} GPUs rank r continues rank r-1's running sums: {k * ld * 4 / 1e6:.0f} MB per hop, {world - 1} hops per pass "
                              f"(= {(world - 1) * k * ld * 4 / 153e9 * 1e3:.1f} ms of xGMI link time at 153 GB/s, serial), + a {k * ld * 4 / 1e6:.0f} MB broadcast",
                "list_len": {"min": int(lens.min()), "mean": round(float(lens.mean()), 1), "max": int(lens.max()), "empty": int((lens == 0).sum())}})
    log(f"[cfg5_rank] {n} x {d} rows, k={k}: build {t_build:.1f} s, {passes:.0f} assign passes at {out['seconds_per_assign_pass']} s = "
        f"{out['assign_pass_algorithmic_tflops']} algorithmic TFLOP/s (contraction alone {out['contraction_algorithmic_tflops']}); "
        f"library peak {_gb(mem_peak - mem0)} GB = {out['peak_over_rows_bytes']} x the rows")
    # properties the reference's build guarantees (ivfflat.rs:123-127): every vec id in exactly one list; assignments <-> lists
    asg = ix.local_assignments
    ok_len = bool(np.array_equal(np.bincount(asg.astype(np.int64), minlength=k).astype(np.uint64), lens))
    # self-retrieval at distance exactly 0.0 (assign and search use the same bit-symmetric D and first-min rule: SURVEY.md 8c)
    probe_ids = np.linspace(0, n - 1, 64).astype(np.int64)
    Qs = X[torch.from_numpy(probe_ids).to(dev)].contiguous()
    oi = torch.zeros(64, 1, dtype=torch.int64, device=dev); od = torch.ones(64, 1, device=dev); oc = torch.zeros(64, dtype=torch.int32, device=dev)
    ix.search_dev(Qs.data_ptr(), ld, 64, 1, 0, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    torch.cuda.synchronize(); ix.poll()
    d0 = od.cpu().numpy()[:, 0]; i0 = oi.cpu().numpy()[:, 0]
    # (a duplicate row with a lower vec id may win the tie: the distance must be exactly 0.0, the id's row must be bit-equal)
    same = all(bool(torch.equal(X[int(i0[j])], X[int(probe_ids[j])])) for j in range(64))
    out["properties"] = {"assignments_match_list_lengths": ok_len, "self_retrieval_distance_exactly_zero": bool(np.all(d0 == 0.0) and same), "queries": 64}
    log(f"[cfg5_rank] properties: {out['properties']}")
    ix.close()
    del X
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    kv = dict(a.split("=", 1) for a in sys.argv[2:])
    gi = lambda k_, dflt: int(kv.get(k_, os.environ.get(k_, dflt)))
    if which == "cfg4":
        res = cfg4_rank(rows=gi("ROWS", 100_000_000), nlist=gi("NLIST", 16384), rank=gi("RANK", 0), world=gi("WORLD", 8), chunk=gi("CHUNK", 2_000_000),
                        sample=gi("SAMPLE", 10_000_000), check=gi("CHECK", 32), steps=gi("STEPS", 20), iters=gi("ITERS", 2), d=gi("DIM", 768))
    else:
        res = cfg5_rank(rows_total=gi("ROWS", 50_000_000), world=gi("WORLD", 8), k=gi("K", 65536), iters=gi("ITERS", 1), d=gi("DIM", 768))
    path = kv.get("OUT", os.environ.get("OUT", f"gpurun_out/rank_nominal_{which}.json"))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))
