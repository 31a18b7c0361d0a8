#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config:
queries/sec + recall@10, IVFFlat N=10M d=768 nlist=4096 nprobe=32, batch=1024 queries.

A "step" = one pass of the hot path over one batch of 1024 synthetic queries:
vers_ivf_search_dev (coarse quantiser -> plan -> inverted-list scan -> merge), queries and
outputs resident in HBM.  The corpus is synthetic (tests/datagen.py "Dist-C": clustered unit
vectors), generated in HBM by vers_gen_rows_dev; the index is built on the device by
vers_ivf_build_dev (k-means: attempts=1, --kmeans-iters iterations) before the timed region.

Prints ONE JSON line (rank 0).  Extra keys: `roofline` (dominant kernel = the inverted-list scan,
algorithmic bytes = bytes of the union of probed lists + centroids), `cpu_baseline` (the C
restatement of the reference path, oracle/vers_oracle.c, timed on one host core on a bounded
sample of the same queries), `recall_at_10`.

Multi-GPU (`python bench.py --gpus N`, which starts its own N ranks, or `torchrun ... bench.py --gpus N`): strong
scaling -- the SAME corpus and the same query batches.  No rank ever holds the corpus: rank r generates only rows
[r*N/W, (r+1)*N/W), build_index runs row-sharded (vers_ivf_build_sharded_dev: local assign, chained exact centroid
sums, rows shipped to the owners of their lists with one all-to-all), the inverted lists end up sharded by cluster
(LPT by list length); per batch every rank runs the replicated coarse quantiser and scans only its lists, partial
top-k are exchanged with ONE RCCL all-gather and merged on every rank.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


from benchlib.line import LINE_LIMIT, compact_line, emit  # noqa: E402,F401  (the ONE stdout line: contract keys + numbers; everything else to a side file / stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extra-file", default=None, help="where the FULL result (every leg with its prose) is written; default gpurun_out/bench_extra_n<gpus>.json")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus size N")
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--kmeans-iters", type=int, default=4)
    ap.add_argument("--modes-per-list", type=int, default=16)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed extra measurements (single query, coarse GEMM, flat cfg2, Dist-U recall)")
    ap.add_argument("--no-nominal", action="store_true", help="skip extra.cfg4_rank / extra.cfg5_rank (one rank's nominal share of the two 8-GPU configs on this GPU; ~1.5 min)")
    ap.add_argument("--f32-rows", action="store_true", help="no fp16 shadow: the f32 rows feed the batched list scan (round 1's configuration; same as VERS_SHADOW=0)")
    ap.add_argument("--streams", type=int, default=3, help="batches in flight: step i is queued on stream i %% S with its own outputs and workspace, so "
                    "the small latency-bound kernels of one batch (coarse quantiser, planning, exact finish) run under the list scan of "
                    "another; every step's work is inside the timed region (1 = strictly one batch after the other)")
    ap.add_argument("--ahead", action="store_true", help="compute the next batch's coarse quantiser on a side stream under the current "
                    "list scan (vers_ivf_coarse_ahead_dev; same-box A/B at cfg3: +0.8 %% -- the scan already fills the chip)")
    args = ap.parse_args()
    if args.f32_rows:
        os.environ["VERS_SHADOW"] = "0"   # (read once by the library when it loads; the ranks inherit it)

    # `python bench.py --gpus N` outside a launcher: this process becomes the parent of N ranks.  It has not touched
    # the GPU (torch is not even imported yet) and never will; it relays rank 0's JSON line and the ranks' status.
    from vers_amd.launch import spawn_ranks, under_launcher
    if args.gpus > 1 and not under_launcher():
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        log(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}")
    # VERS_BENCH_FORCE_SHARDED=1: the multi-rank code path with however many ranks there are -- with ONE rank it is the smoke test
    # of everything `--gpus N` runs on the GPUs (nccl process group, row-sharded build entry, libvers_rccl.so's communicator made
    # from an id broadcast through torch, vers_ivf_search_sharded_dev with ncclAllGather on the batch's stream) minus the peers
    multi = world > 1 or os.environ.get("VERS_BENCH_FORCE_SHARDED") == "1"
    if multi and world == 1 and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    # stdout carries ONE JSON line.  RCCL prints its start-up banner to stdout under NCCL_DEBUG=VERSION (this pool's default): the
    # adapter reports the same facts on stderr (vers_rccl_versions), so that level is dropped; any other level's output goes to stderr
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    import torch
    # one process per GPU.  (VERS_BENCH_BACKEND=gloo + fewer GPUs than ranks is a debugging aid only: it lets
    # the multi-rank code path run on a 1-GPU box, staging the all-gather through host memory.)
    backend = os.environ.get("VERS_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    dist = None
    if multi:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from tests import datagen as dg
    from vers_amd import capi
    from vers_amd.index import IVFFlatIndex

    n, d, nlist, nprobe, B, top_k = args.rows, args.d, args.nlist, args.nprobe, args.batch, args.top_k
    ld = (d + 3) // 4 * 4
    # Dist-C with 16 modes per list.  (SURVEY.md 8d suggested nlist/4 modes; measured on MI355X that
    # degenerates: 4 iterations of k-means leave 3/4 of the lists empty and the rest at 4x the mean, so
    # a query would probe ~1 real list instead of nprobe.  16 modes/list gives balanced lists of ~N/nlist.)
    n_modes = max(1, args.modes_per_list * nlist)
    sigma = float(dg.default_sigma(d))
    SEED_X, SEED_Q, SEED_C = 0x5EED0001, 0x5EED0002, 0x5EEDC0DE

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- corpus + index (not timed) -----------------------------------------------------------------
    # rank r generates and holds ONLY rows [lo, hi) (the generator is counter-based: any row on any GPU)
    lo, hi = rank * n // world, (rank + 1) * n // world
    t0 = time.perf_counter()
    X = torch.empty(hi - lo, ld, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(X.data_ptr(), hi - lo, d, ld, 1, SEED_X, SEED_C, n_modes, sigma, start_row=lo)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    init = (dg.mix64(np.uint64(0xB01D) + np.arange(nlist, dtype=np.uint64)) % np.uint64(n)).astype(np.uint64)
    index = IVFFlatIndex(d, device=dev_index)
    capi.mem_stats(reset_peak=True)
    capi.build_stats(reset=True)
    t0 = time.perf_counter()
    comm = None
    # On the GPUs (backend nccl = RCCL) the library's OWN adapter drives every exchange of the run: libvers_rccl.so's
    # vers_rccl_comm callbacks for the row-sharded build (what a Rust host would link), its vers_rccl_gather for the search.
    # VERS_BENCH_BUILD_COMM=torch puts vers_amd.dist.TorchComm (Python callbacks over torch.distributed) behind the build
    # instead -- it counts calls / bytes / seconds per exchange.  RcclComm.from_torch is collective-safe: it raises on every
    # rank or on none.
    rccl_comm = None
    if multi and backend == "nccl" and os.environ.get("VERS_BENCH_EXCHANGE", "rccl") == "rccl":
        try:
            from vers_amd import rccl as vrccl
            rccl_comm = vrccl.RcclComm.from_torch(dev_index)
            if rank == 0:
                log(f"[bench] libvers_rccl.so: {vrccl.versions()}")
        except Exception as e:  # (the library or RCCL refused on some rank: every rank is here)
            log(f"[bench] rank {rank}: RCCL adapter unavailable ({e!r}); torch.distributed carries the exchanges")
    if multi:
        from vers_amd.dist import TorchComm
        build_comm_kind = os.environ.get("VERS_BENCH_BUILD_COMM", "rccl" if rccl_comm is not None else "torch")
        comm = rccl_comm if (build_comm_kind == "rccl" and rccl_comm is not None) else TorchComm(device=dev_index)
        kept = index.build_sharded_dev(X.data_ptr(), hi - lo, ld, lo, n, nlist, 1, args.kmeans_iters, init, comm)
    else:
        kept = index.build_dev(X.data_ptr(), n, nlist, 1, args.kmeans_iters, init)
    t_build = time.perf_counter() - t0
    assert kept
    bph = capi.build_phases()           # (before the reset below: the phases and the event-timed stats share one reset)
    bst = capi.build_stats(reset=True)
    del X
    torch.cuda.empty_cache()
    lens = index.list_lengths()
    mem_now, mem_peak = capi.mem_stats()
    if rank == 0:
        import zlib
        fp = zlib.crc32(np.ascontiguousarray(index.get_centroids()).tobytes(), zlib.crc32(np.ascontiguousarray(lens).tobytes()))
        mp, mf = capi.assign_stats()
        log(f"[bench] corpus {n}x{d} generated in {t_gen:.1f}s; build_index (k-means {int(index.iterations[0])} iters + final assign) "
            f"{t_build:.1f}s; cost {float(index.cost):.1f} (bits {np.float32(index.cost).view(np.uint32):#010x}); "
            f"list len min/mean/max {int(lens.min())}/{lens.mean():.0f}/{int(lens.max())}; "
            f"index fingerprint (centroid bits + list lengths) {fp:#010x}; matrix-core assign: {mp} points, {mf} re-done exactly")
    sharded_build = None
    if multi:
        # every rank's peak of library memory during the row-sharded build (its rows are the caller's: generated above)
        pk = torch.tensor([float(mem_peak), float(mem_now), float(hi - lo) * ld * 4.0], device=dev, dtype=torch.float64)
        allpk = [torch.zeros_like(pk) for _ in range(world)]
        if backend == "nccl":
            dist.all_gather(allpk, pk)
        else:
            hl = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(hl, pk.cpu()); allpk = hl
        if rank == 0:
            sharded_build = {"rows_per_rank": hi - lo, "rank_rows_gb": round((hi - lo) * ld * 4 / 1e9, 2), "whole_corpus_gb": round(n * d * 4 / 1e9, 2),
                             "library_mem_peak_gb_per_rank": [round(float(t[0]) / 1e9, 2) for t in allpk],
                             "library_mem_after_build_gb_per_rank": [round(float(t[1]) / 1e9, 2) for t in allpk],
                             "peak_over_rank_rows": round(max(float(t[0]) / max(1.0, float(t[2])) for t in allpk), 3),
                             "build_index_s": round(t_build, 2), "backend": backend,
                             "exchanges_through": "libvers_rccl.so vers_rccl_comm (native callbacks on the adapter's stream)" if comm is rccl_comm else "vers_amd.dist.TorchComm (Python callbacks over torch.distributed)"}
            if hasattr(comm, "calls"):   # (TorchComm counts; the native callbacks do not: VERS_BENCH_BUILD_COMM=torch for the counters)
                sharded_build.update({"exchange_calls": dict(comm.calls), "exchange_bytes": dict(comm.bytes),
                                      "exchange_seconds_rank0": {k_: round(v_, 3) for k_, v_ in comm.seconds.items()},
                                      "chain_hop_ms": round(1e3 * (comm.seconds["send"] + comm.seconds["recv"]) / max(1, comm.calls["send"] + comm.calls["recv"]), 2)})
    if multi and rank == 0:
        own = index.owners()
        log(f"[bench] row-sharded build over {world} ranks: {hi - lo} rows generated per rank, library device memory peak "
            f"{mem_peak / 1e9:.2f} GB / now {mem_now / 1e9:.2f} GB on rank 0 (whole corpus: {n * d * 4 / 1e9:.2f} GB); "
            f"exchanges through {'libvers_rccl.so (vers_rccl_comm)' if comm is rccl_comm else 'TorchComm: calls ' + str(comm.calls) + ', bytes ' + str(comm.bytes)}")
        log(f"[bench] lists sharded over {world} ranks (LPT): rows per rank "
            f"{[int(lens[own == r].sum()) for r in range(world)]}")

    # ---- queries: distinct batches drawn from the same distribution (not from the corpus) -------------
    n_batches = max(1, min(args.steps + args.warmup, 8))
    Q = torch.empty(n_batches * B, ld, dtype=torch.float32, device=dev)
    capi.gen_rows_dev(Q.data_ptr(), n_batches * B, d, ld, 1, SEED_Q, SEED_C, n_modes, sigma)
    # S batches in flight: step i runs on stream i % S with its own outputs; the library gives a stream's calls their own
    # workspace.  The list scan fills every CU, but what surrounds it -- coarse quantiser, planning, exact finish: a dozen
    # latency-bound launches that do NOT shrink when the lists are sharded over GPUs -- overlaps with another batch's scan.
    # (Round 2 measured this and backed off because the GPU hung in some stream mixes -- its planning kernel spun on a grid
    # barrier; nothing in the search path waits for another block any more: DESIGN.md section 5.)
    S = max(1, args.streams)
    streams = [torch.cuda.current_stream()] if S == 1 else [torch.cuda.Stream() for _ in range(S)]
    outs = [dict(ids=torch.zeros(B, top_k, dtype=torch.int64, device=dev), dst=torch.zeros(B, top_k, dtype=torch.float32, device=dev),
                 cnt=torch.zeros(B, dtype=torch.int32, device=dev),
                 part=torch.empty(2, B, top_k, dtype=torch.int64, device=dev),           # [keys | vec ids] of this rank
                 allp=torch.empty(world, 2, B, top_k, dtype=torch.int64, device=dev)) for _ in range(S)]
    ids, dst, cnt = outs[0]["ids"], outs[0]["dst"], outs[0]["cnt"]
    st = torch.cuda.current_stream().cuda_stream   # the untimed legs below run one call at a time on the default stream
    torch.cuda.synchronize()

    # The ONE collective per batch.  Default on the GPUs: the library drives RCCL itself (libvers_rccl.so: ncclAllGather queued on
    # the batch's own stream inside vers_ivf_search_sharded_dev -- no host synchronisation, no second stream, no Python between
    # the partial search and the merge).  VERS_BENCH_EXCHANGE=torch: the round-3 path (torch.distributed all_gather_into_tensor
    # between two library calls).  gloo (debugging on fewer GPUs than ranks): the same entry point with gloo behind vers_gather_t.
    gather, exchange_kind, exchange_tag = None, None, None
    if multi:
        want = os.environ.get("VERS_BENCH_EXCHANGE", "rccl" if backend == "nccl" else "gloo")
        if want == "rccl" and backend == "nccl":
            gather = rccl_comm   # (made before the build, the same on every rank or on none: RcclComm.from_torch)
            if gather is not None:
                exchange_kind = "libvers_rccl.so: ncclAllGather on the batch's stream (vers_ivf_search_sharded_dev)"
                exchange_tag = "rccl_allgather_inside_search_sharded_dev"
        elif want == "gloo" or backend != "nccl":
            from vers_amd.dist import TorchGather
            gather = TorchGather(device=dev_index)
            exchange_kind = f"vers_gather_t over torch.distributed/{backend} staged through host memory (vers_ivf_search_sharded_dev)"
            exchange_tag = f"{backend}_gather_inside_search_sharded_dev"
        if gather is None:
            exchange_kind = "torch.distributed all_gather_into_tensor between vers_ivf_search_partial_dev and vers_topk_merge_dev"
            exchange_tag = "torch_allgather_between_partial_and_merge"

    def exchange(o):
        dist.all_gather_into_tensor(o["allp"], o["part"])                   # (the torch path: RCCL through ProcessGroupNCCL)

    def step(i, S=S):
        qb = Q[(i % n_batches) * B:]
        o = outs[i % S]
        with torch.cuda.stream(streams[i % S]):
            sh = streams[i % S].cuda_stream
            if args.ahead:  # the NEXT batch's coarse quantiser runs on a side stream under this batch's list scan
                index.coarse_ahead_dev(Q[((i + 1) % n_batches) * B:].data_ptr(), ld, B, nprobe, sh)
            if not multi:
                index.search_dev(qb.data_ptr(), ld, B, top_k, nprobe, o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(), sh)
            elif gather is not None:
                index.search_sharded_dev(gather.gather_ptr() if hasattr(gather, "gather_ptr") else gather.ptr(), qb.data_ptr(), ld, B, top_k, nprobe,
                                         o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(), sh)
            else:
                index.search_partial_dev(qb.data_ptr(), ld, B, top_k, nprobe, o["part"][0].data_ptr(), o["part"][1].data_ptr(), sh)
                exchange(o)
                IVFFlatIndex.merge_partials_dev(o["allp"].data_ptr(), o["allp"].data_ptr() + 8 * B * top_k, 2 * B * top_k, world, B, top_k,
                                                nprobe, o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(), sh)

    if B == 1:  # (--batch 1: the run's roofline comes from the event records single-query calls do not make unless asked)
        capi.set_option("scan_events", 1)
    for i in range(args.warmup):
        step(i)
    for x in streams:
        index.poll(x.cuda_stream)
    index.scan_times(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    for x in streams:
        index.poll(x.cuda_stream)  # NaN / insufficient latch -> raises
    last_o = outs[(args.warmup + args.steps - 1) % S]   # the results of the last timed batch (recall, CPU comparison)
    ids, dst, cnt = last_o["ids"], last_o["dst"], last_o["cnt"]
    ls = index.last_scan()
    scan_ms = index.scan_times(reset=True)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = args.steps * B / elapsed
    scan_ms_one = None
    if multi and S > 1:
        # every rank: the same steps one batch after the other -- the dominant kernel's duration without the queueing behind other
        # batches' scans that an event pair measures when several are in flight (see the one-GPU leg below)
        for i in range(args.warmup):
            step(i, 1)
        index.poll(streams[0].cuda_stream)
        index.scan_times(reset=True)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, 1)
        barrier()
        t1 = time.perf_counter() - t1
        index.poll(streams[0].cuda_stream)
        scan_ms_one = (index.scan_times(reset=True), t1)

    # ---- roofline of the dominant kernel (inverted-list scan), last batch's geometry -----------------
    # Algorithmic bytes = what the kernel's algorithm has to read: the rows of the union of probed lists in the form the
    # kernel streams them -- the fp16 shadow (2 B per element + 4 B of |x|^2 per row) by default, the f32 rows with
    # VERS_SHADOW=0 / on the ordered-chain path -- plus the centroids (SURVEY 8d counts them with the batch).  The f32-row
    # figure of SURVEY 8d for the same launch is reported next to it (`f32_rows_equivalent`): the shadow kernel beats that
    # roofline because it does not read those bytes, not because it streams faster than HBM.
    scan_mean_ms = float(np.mean(scan_ms)) if len(scan_ms) else float("nan")
    pst = index.prescan_stats()
    mfma_scan = pst["batches"] > 0
    shadow = mfma_scan and index.shadow_state()["active"]
    f32_bytes = ls["union_rows"] * d * 4 + nlist * d * 4
    algo_bytes = ls["union_rows"] * (d * 2 + 4) + nlist * d * 4 if shadow else f32_bytes
    streamed = ls["streamed_rows"] * (d * 2 + 4) if shadow else ls["streamed_rows"] * d * 4
    achieved = algo_bytes / (scan_mean_ms * 1e-3) / 1e9
    # HBM traffic of the same kernel from the PMC passes of the committed rocprofv3 run (bench.py cannot collect
    # counters itself); used only when it was measured on this exact configuration and kernel.
    kernel_id = ("prescan_kernel_g<true" if shadow else "prescan_kernel_g<false") if mfma_scan else "scan_kernel"
    # (which instantiation ran follows the planner's rule, ivf_plan.hip: 64-query blocks with the query block as fp16 hi only where they
    # fit LDS -- d <= 960 --, else 32 queries hi + lo up to d = 1152, 32 hi-only up to 2304, 16 hi-only up to 4608; option pre_wide=0: no 64)
    wide64 = capi.env_option("pre_wide", 1) != 0 and capi.env_option("pre_narrow", 0) == 0 and ((d + 63) // 64 * 64) * 64 * 2 + 16 + 64 * 64 * 8 + 6 * 64 * 4 <= 160 * 1024
    pre_inst = ("prescan_kernel_g<true, 64, IvfSrc<64>, false> (64 queries per block as two sets of 32, query block fp16 hi only" if wide64 else
                "prescan_kernel_g<true, 32, IvfSrc<32>, true> (32 queries per block, query block fp16 hi + lo" if d <= 1152 else
                "prescan_kernel_g<true, 32, IvfSrc<32>, false> (32 queries per block, query block fp16 hi only")
    if B == 1 and shadow:   # (--batch 1: the single query's own scan of the shadow)
        kernel_id = "scan1h_kernel"
    kernel_name = ("scan1h_kernel (one query on the fp16 shadow: a 64-row tile per wave, VALU dot products; exact f32 finish in ivf_rescore_kernel<16>)" if B == 1 and shadow else
                   pre_inst + "; inverted-list scan: fp16 shadow rows -> v_mfma_f32_32x32x16_f16; exact f32 finish in ivf_rescore_kernel)" if shadow
                   else "prescan_kernel_g<false, 32, IvfSrc<32>> (inverted-list scan on the f32 matrix cores; exact finish in ivf_rescore_kernel)" if mfma_scan
                   else "scan_kernel<QG,0,IvfSrc<QG>> (inverted-list scan, ordered f32 chains; QG = 16 at this shape)")
    traffic, traffic_source = None, None
    for tf in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):  # newest PMC run of this exact configuration
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            if (tj["config"] == {"rows": n, "d": d, "nlist": nlist, "nprobe": nprobe, "batch": B} and not multi
                    and tj["kernel"].startswith(kernel_id)):
                traffic = tj["hbm_read_bytes_per_launch"]
                traffic_source = f"profiles/{tf}: rocprofv3 --pmc FETCH_SIZE pass of this configuration (committed; not collected in this run)"
                break
        except (OSError, KeyError, ValueError):
            pass
    if rank == 0:
        log(f"[bench] list scan on the matrix cores: {pst['batches']} batches, {pst['fallback_queries']} queries failed the certificate "
            f"and were re-scanned exactly; fp16 shadow rows {'in use' if shadow else 'not in use'}")
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(algo_bytes), "streamed_bytes_per_launch": int(streamed),
                "row_operand": "fp16 shadow of the rows, 2 B/element (results are the exact f32 bits)" if shadow else "f32 rows",
                "launch_ms": round(scan_mean_ms, 4), "launches_timed": int(len(scan_ms)), "work_items": int(ls["items"])}
    if shadow:
        roofline["f32_rows_equivalent"] = {"bytes_per_launch": int(f32_bytes), "GBs": round(f32_bytes / (scan_mean_ms * 1e-3) / 1e9, 1),
                                           "frac_of_peak": round(f32_bytes / (scan_mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "note": "SURVEY 8d's f32 bytes of the same union of lists over this launch time: above 1 because the kernel reads the half-size shadow instead"}

    # ---- recall@10 against the exact scan (utils::search_exhaustive over the same values) --------------
    last = args.warmup + args.steps - 1
    recall = None
    ids_h = ids.cpu().numpy().astype(np.uint64); dst_h = dst.cpu().numpy(); cnt_h = cnt.cpu().numpy()
    if not args.no_recall:
        nq_r = min(B, 1024)
        eids = torch.zeros(nq_r, top_k, dtype=torch.int64, device=dev)
        edst = torch.zeros(nq_r, top_k, dtype=torch.float32, device=dev)
        ecnt = torch.zeros(nq_r, dtype=torch.int32, device=dev)
        qb = Q[(last % n_batches) * B:]
        t0 = time.perf_counter()
        index.search_exhaustive_dev(qb.data_ptr(), ld, nq_r, top_k, 0, eids.data_ptr(), edst.data_ptr(), ecnt.data_ptr(), st)
        index.poll(st)
        t_ex = time.perf_counter() - t0
        e = eids.cpu().numpy().astype(np.uint64)
        if multi:  # each rank scanned only its rows: combine the per-rank exact top-k by (distance, vec id)
            ed = edst.cpu().numpy(); ec = ecnt.cpu().numpy()
            ed[np.arange(top_k)[None, :] >= ec[:, None]] = np.inf
            gl = [None] * world
            dist.all_gather_object(gl, (e, ed))
            ae = np.concatenate([g[0] for g in gl], axis=1); ad = np.concatenate([g[1] for g in gl], axis=1)
            order = np.lexsort((ae, ad), axis=1)[:, :top_k]
            e = np.take_along_axis(ae, order, axis=1)
        hits = sum(len(set(ids_h[q, :cnt_h[q]].tolist()) & set(e[q].tolist())) for q in range(nq_r))
        recall = hits / float(nq_r * top_k)
        if rank == 0:
            log(f"[bench] recall@{top_k} = {recall:.4f} over {nq_r} queries (exact scan took {t_ex:.2f}s)")

    # ---- full-size property: self-retrieval -------------------------------------------------------------------
    # a stored row queried bit-identically comes back first at distance exactly 0.0 (assign and search use the same
    # symmetric ordered distance and the same first-minimum rule: ivfflat.rs:36-43 vs :159-160)
    self_ok = None
    if not multi:
        own = [c for c in range(0, nlist, max(1, nlist // 8)) if lens[c] > 0][:8]
        rows_ids = [index.get_list(c) for c in own]
        sq = np.stack([r[0][len(r[1]) // 2] for r in rows_ids]); sid = np.array([r[1][len(r[1]) // 2] for r in rows_ids])
        sqd = torch.from_numpy(sq).to(dev)
        sids = torch.zeros(len(own), top_k, dtype=torch.int64, device=dev); sdst = torch.zeros(len(own), top_k, device=dev)
        scnt = torch.zeros(len(own), dtype=torch.int32, device=dev)
        index.search_dev(sqd.data_ptr(), d, len(own), top_k, nprobe, sids.data_ptr(), sdst.data_ptr(), scnt.data_ptr(), st)
        index.poll(st)
        self_ok = bool((sids[:, 0].cpu().numpy().astype(np.uint64) == sid).all() and (sdst[:, 0].cpu().numpy() == 0.0).all())
        log(f"[bench] self-retrieval of {len(own)} stored rows at N={n}: {'ok' if self_ok else 'FAILED'}")

    # ---- extra measurements, outside the timed region (north_star's other targets) ---------------
    extra = {}
    MFMA_F32_PEAK_TF = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
    BF16_DENSE_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_bf16 / _f16 dense peak (same guide); the assign cascade's first filter spends ONE fp16 product per f32 one

    def assign_entry(bs, shape_pts, k_, wall_s, note):
        """k-means assign (ivfflat.rs:29-46) as the build just ran it: the contraction launches by HIP events (vers_build_stats)"""
        if bs["gemm_launches"] <= 0 or bs["gemm_ms"] <= 0:
            return None
        tf = bs["gemm_flop"] / (bs["gemm_ms"] * 1e-3) / 1e12
        e = {"kernel": "the assign cascade's first filter (round 6): 256 x 256 block tiles of points x centroids, ONE v_mfma_f32_32x32x16_f16 per k-step on fp16 operands, certified with their measured residuals; arg-min fused into the epilogue; the open points through assign_tile_rescan_kernel. "
                       "Below 4096 centroids dist_gemm_x3w_kernel<2, 1> (register staged, converts the f32 batch while it stages it); from 4096 on dist_gemm_h_kernel (persistent, both operands fp16 in memory -- the batch's conversion pass is inside the timed launches --, 64-column K-tiles by LDS-DMA in whole cache lines). "
                       "dist_gemm_x3w_kernel<2, 3> -- three bf16 products of hi/lo-split operands -- when the probe says the cascade does not pay or vers_set_option('assign_terms', 3)",
             "shape": [int(shape_pts), int(k_), d], "launches": int(bs["gemm_launches"]), "us_per_launch": round(bs["gemm_ms"] / bs["gemm_launches"] * 1e3, 1),
             "algorithmic_tflops": round(tf, 1), "frac_of_f16_dense": round(tf / BF16_DENSE_PEAK_TF, 4), "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 4),
             "assign_pass_ms": round(bs["assign_ms"] / max(1.0, bs["assign_passes"]), 2), "assign_passes": int(bs["assign_passes"]),
             "points_redone_exactly_pct": round(100.0 * bs["redone_points"] / max(1.0, bs["gemm_flop"] / (2.0 * k_ * d)), 3),
             "update_centroids_ms_total": round(bs["update_ms"], 2), "cost_fold_ms_total": round(bs["cost_ms"], 2), "build_index_s": round(wall_s, 3), "note": note}
        for pf in ("r06_kmeans.json",):  # MFMA-busy of this kernel from the committed PMC pass (bench.py cannot collect counters itself)
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", pf)))
                e["mfma_busy_pct"] = pj["mfma_busy_pct"].get(str(int(k_)))
                e["mfma_busy_source"] = f"profiles/{pf}: rocprofv3 --pmc pass (committed; not collected in this run)"
                break
            except (OSError, KeyError, ValueError, AttributeError):
                pass
        return e

    if rank == 0:
        sst = index.shadow_state()
        lay = index.layout_bytes()
        hbm_total = int(torch.cuda.get_device_properties(dev).total_memory)
        rows_stored = max(1, n if not multi else int(lens[index.owners() == rank].sum()))
        per_row = float(mem_now) / rows_stored
        per_row_noshadow = float(mem_now - sst["bytes"] - lay["rowmajor"]) / rows_stored
        per_row_shadow_only = float(mem_now - lay["rowmajor"]) / rows_stored
        extra["memory"] = {"library_bytes_now": int(mem_now), "library_bytes_peak_during_build": int(mem_peak), "f32_tile_rows_bytes": int(lay["rows"]),
                           "shadow_bytes": int(sst["bytes"]), "shadow_active": bool(sst["active"]), "rowmajor_copy_bytes": int(lay["rowmajor"]), "rowmajor_kept": bool(lay["rowmajor"] > 0),
                           "bytes_per_stored_row": round(per_row, 1), "bytes_per_stored_row_without_rowmajor_copy": round(per_row_shadow_only, 1),
                           "bytes_per_stored_row_without_shadow_and_rowmajor_copy": round(per_row_noshadow, 1), "hbm_bytes": hbm_total,
                           "max_N_per_gpu_as_configured": int((hbm_total - (8 << 30)) / per_row), "max_N_per_gpu_with_shadow": int((hbm_total - (8 << 30)) / per_row_shadow_only),
                           "max_N_per_gpu_without_shadow": int((hbm_total - (8 << 30)) / per_row_noshadow),
                           "note": "rows incl. list slack + row ids + |x|^2, + the fp16 shadow the batched scan streams, + the row-major f32 copy the exact finish gathers "
                                   "from (optional: kept while the rows take <= 1/4 of the device, VERS_ROWMAJOR=0 drops it); 8 GB set aside for per-batch scratch and the caller"}
        extra["build_index_s"] = round(t_build, 3)
        extra["build_index_phases_ms"] = dict(bph, note="host wall clock per phase of this build (vers_build_phases); assign_first_pass_ms pays the process's cold start "
                                                        "(code load, first launches, first touch of the 10s of GB just allocated); install_lists = grouping + storage allocation + row "
                                                        "placement; derive = centroid operands, |x|^2, fp16 shadow, row-major copy")
        if sharded_build:
            extra["sharded_build"] = sharded_build
        ke = assign_entry(bst, min(131072, hi - lo), nlist, t_build, f"the timed index's own build: N={n} over {world} rank(s), k={nlist}, {int(index.iterations[0])} iterations + final assign")
        if ke:
            extra["kmeans_assign"] = ke
            log(f"[bench] k-means assign contraction [{ke['shape'][0]}x{nlist}x{d}]: {ke['us_per_launch']} us per launch = {ke['algorithmic_tflops']} algorithmic TFLOP/s "
                f"({ke['points_redone_exactly_pct']} % of the points re-done exactly); update {ke['update_centroids_ms_total']} ms, cost fold {ke['cost_fold_ms_total']} ms in a {t_build:.2f} s build")
    if rank == 0 and not multi and not args.no_extra and d == 768:
        # cfg5's cluster count on one GPU: N = 1M, k = 65536, one iteration + the final assign (2 passes)
        nk, kk = 1_048_576, 65536
        Xk = torch.empty(nk, ld, dtype=torch.float32, device=dev)
        capi.gen_rows_dev(Xk.data_ptr(), nk, d, ld, 1, SEED_X + 0x200, SEED_C, n_modes, sigma)
        ik = IVFFlatIndex(d, device=dev_index)
        initk = (dg.mix64(np.uint64(0xB16C) + np.arange(kk, dtype=np.uint64)) % np.uint64(nk)).astype(np.uint64)
        capi.build_stats(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ik.build_dev(Xk.data_ptr(), nk, kk, 1, 1, initk)
        tk = time.perf_counter() - t0
        bk = capi.build_stats(reset=True)
        ik.close(); del Xk
        torch.cuda.empty_cache()
        ke = assign_entry(bk, min(131072, nk), kk, tk, f"BASELINE.json cfg5's cluster count on one GPU: N={nk}, k={kk}, 1 iteration + final assign")
        if ke:
            extra["kmeans_assign_k65536"] = ke
            log(f"[bench] k-means at k=65536 (N={nk}): build {tk:.2f} s, contraction {ke['algorithmic_tflops']} algorithmic TFLOP/s, {ke['points_redone_exactly_pct']} % re-done exactly")
    if rank == 0:
        # the batched coarse quantiser's contraction, queries x centroids [B x d].[d x nlist]: the production kernel (three bf16
        # MFMA products of hi/lo-split operands) as timed in the last step, and the f32 MFMA kernel on the same batch (same
        # results: both are pre-filters behind the exact re-score + certificate) -- north_star's "MFMA utilisation on the
        # batch-1024 query GEMM" is priced on the f32 kernel against the f32 MFMA peak
        def coarse_entry(kernel, cm):
            tf = 2.0 * B * nlist * d / (cm["gemm_ms"] * 1e-3) / 1e12
            return {"kernel": kernel, "shape": [B, nlist, d], "us": round(cm["gemm_ms"] * 1e3, 1), "algorithmic_tflops": round(tf, 1),
                    "select_rescore_us": round(cm["select_ms"] * 1e3, 1)}
        try:
            # (the legs above left other calls -- single queries, exact scans -- as the handle's last: a few batches again; with the
            # lists sharded, this rank's part of the search: no collective)
            for i in range(8):
                if not multi:
                    index.search_dev(Q[(i % n_batches) * B:].data_ptr(), ld, B, top_k, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
                else:
                    index.search_partial_dev(Q[(i % n_batches) * B:].data_ptr(), ld, B, top_k, nprobe, outs[0]["part"][0].data_ptr(),
                                             outs[0]["part"][1].data_ptr(), st)
            index.poll(st)
            extra["coarse_gemm"] = coarse_entry("dist_gemm_x3_kernel<false> (3 x v_mfma_f32_32x32x16_bf16 on hi/lo-split operands, 128x128 block tiles)",
                                                index.last_coarse_ms())
            if not multi and not args.no_extra:
                capi.set_option("gemm_x3", 1)
                for i in range(16):  # (the GPU has idled through the recall / CPU legs: a few dozen ms of work bring the clocks back up)
                    index.search_dev(Q[(i % n_batches) * B:].data_ptr(), ld, B, top_k, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
                index.poll(st)
                f32e = coarse_entry("dist_gemm_kernel<false> (v_mfma_f32_32x32x2_f32, 128x128 block tiles)", index.last_coarse_ms())
                f32e["peak_tflops"] = MFMA_F32_PEAK_TF; f32e["frac"] = round(f32e["algorithmic_tflops"] / MFMA_F32_PEAK_TF, 4)
                extra["coarse_gemm_f32"] = f32e
                capi.set_option("gemm_x3", 3)
        except capi.VersError as e:
            capi.set_option("gemm_x3", 3)
            log(f"[bench] coarse contraction timings unavailable: {e}")
    def timed_steps(np_, S=S):
        """the timed region's loop again (same warm-up, same step count, same streams) with another nprobe / row operand"""
        def stp(i):
            o = outs[i % S]
            sh = streams[i % S].cuda_stream
            index.search_dev(Q[(i % n_batches) * B:].data_ptr(), ld, B, top_k, np_, o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(), sh)
        for i in range(args.warmup):
            stp(i)
        for x in streams:
            index.poll(x.cuda_stream)
        index.scan_times(reset=True)
        torch.cuda.synchronize(); t_ = time.perf_counter()
        for i in range(args.steps):
            stp(args.warmup + i)
        torch.cuda.synchronize(); t_ = time.perf_counter() - t_
        for x in streams:
            index.poll(x.cuda_stream)
        return t_, index.scan_times(reset=True)

    if rank == 0 and multi and scan_ms_one is not None and len(scan_ms_one[0]):
        m1 = float(np.mean(scan_ms_one[0]))
        roofline["timed_region"] = {"launch_ms": roofline["launch_ms"], "achieved": roofline["achieved"], "frac": roofline["frac"], "launches_timed": roofline["launches_timed"],
                                    "note": f"event pairs around the launches of the timed region, {S} batches in flight: includes the time a launch waits for the CUs another batch's scan still holds"}
        roofline.update({"launch_ms": round(m1, 4), "achieved": round(algo_bytes / (m1 * 1e-3) / 1e9, 1), "frac": round(algo_bytes / (m1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "launches_timed": int(len(scan_ms_one[0])), "measured_on": f"the same {args.steps} steps, one batch in flight, right after the timed region (rank 0's launches)"})
        roofline["one_batch_in_flight"] = {"whole_step_ms": round(scan_ms_one[1] / args.steps * 1e3, 4), "whole_step_queries_per_sec": round(args.steps * B / scan_ms_one[1], 1)}
    if rank == 0 and not multi and S > 1:
        # With several batches in flight an event pair around a list-scan launch also measures how long the launch QUEUED behind
        # another batch's scan (two scans cannot share the chip: each block takes a whole CU) -- not the kernel.  The roofline of the
        # dominant kernel is therefore taken from the same steps run one batch after the other (--streams 1) right after the
        # timed region, same process, same HIP events; the timed region's own figure stays in the line beside it.
        t1s, ms1s = timed_steps(nprobe, 1)
        try:
            extra["exact_finish_us"] = {"us": round(index.last_finish_ms() * 1e3, 1), "kernel": "ivf_rescore_kernel (merge of the partial lists, certificate, exact re-score, emit)",
                                        "rowmajor_copy": bool(index.layout_bytes()["rowmajor"]), "measured_on": "the last step of the one-batch-in-flight pass (HIP events)"}
        except capi.VersError:
            pass
        if len(ms1s):
            m1 = float(np.mean(ms1s))
            roofline["timed_region"] = {"launch_ms": roofline["launch_ms"], "achieved": roofline["achieved"], "frac": roofline["frac"], "launches_timed": roofline["launches_timed"],
                                        "note": f"event pairs around the launches of the timed region, {S} batches in flight: includes the time a launch waits for the CUs "
                                                f"another batch's scan still holds"}
            roofline.update({"launch_ms": round(m1, 4), "achieved": round(algo_bytes / (m1 * 1e-3) / 1e9, 1), "frac": round(algo_bytes / (m1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "launches_timed": int(len(ms1s)),
                             "measured_on": f"the same {args.steps} steps, one batch in flight, right after the timed region (same process, HIP events on the launch stream)"})
            if shadow:
                roofline["f32_rows_equivalent"].update({"GBs": round(f32_bytes / (m1 * 1e-3) / 1e9, 1), "frac_of_peak": round(f32_bytes / (m1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            roofline["one_batch_in_flight"] = {"whole_step_ms": round(t1s / args.steps * 1e3, 4), "whole_step_queries_per_sec": round(args.steps * B / t1s, 1),
                                               "note": "the step's latency: the same steps strictly one after the other"}
            log(f"[bench] one batch in flight: {t1s / args.steps * 1e3:.3f} ms per step, list scan {m1:.3f} ms = {roofline['frac']} of peak "
                f"(timed region, {S} in flight: {roofline['timed_region']['launch_ms']} ms per launch incl. queueing)")
    if rank == 0 and not multi and not args.no_extra and shadow:
        # the same steps with the f32 rows feeding the list scan (round 1's kernel; same index, same results): what the shadow
        # buys, and SURVEY 8d's f32-row figure of the headline -- same warm-up, step count and streams as the timed region
        try:
            capi.set_option("shadow", 0)
            t32, ms32 = timed_steps(nprobe)
            t32_1, ms32_1 = timed_steps(nprobe, 1) if S > 1 else (t32, ms32)   # (the kernel with the chip to itself: the headline's convention)
            if len(ms32) and len(ms32_1):
                mq = float(np.mean(ms32))
                m = float(np.mean(ms32_1))
                extra["list_scan_f32_rows"] = {"kernel": "prescan_kernel_g<false, 32, IvfSrc<32>> (f32 rows -> v_mfma_f32_16x16x1_4b_f32)", "launch_ms": round(m, 4),
                                               "algorithmic_bytes_per_launch": int(f32_bytes), "achieved_GBs": round(f32_bytes / (m * 1e-3) / 1e9, 1),
                                               "frac": round(f32_bytes / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "measured_on": "the same steps one batch in flight (like roofline.frac of the headline)",
                                               "timed_region": {"launch_ms": round(mq, 4), "frac": round(f32_bytes / (mq * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                                "note": f"event pairs with {S} batches in flight: includes queueing behind another batch's scan"},
                                               "one_batch_in_flight_whole_step_ms": round(t32_1 / args.steps * 1e3, 4),
                                               "steps": args.steps, "warmup": args.warmup,
                                               "whole_step_ms": round(t32 / args.steps * 1e3, 4), "whole_step_queries_per_sec": round(args.steps * B / t32, 1),
                                               "note": "the same index and batches with vers_set_option('shadow', 0), timed like the headline (after it): round 1's configuration (VERS_SHADOW=0 / --f32-rows makes it the whole run)"}
        finally:
            capi.set_option("shadow", 1)
    if rank == 0 and not multi and not args.no_extra and nprobe != 0:
        # the reference's OWN mode (nprobe = 0: nearest list, spill while short, ivfflat.rs:166-195) on the same index and batches
        tr, msr = timed_steps(0)
        n1r = 100
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n1r):
            index.search_dev(Q[(i % (n_batches * B)):].data_ptr(), ld, 1, top_k, 0, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
        torch.cuda.synchronize(); e1r = (time.perf_counter() - t0) / n1r
        index.poll(st)
        # ... and the trait call itself in that mode: vers_ivf_search, host pointers, one query per call (Index::search_approximate)
        import ctypes as Cr_
        qhr = np.ascontiguousarray(Q[:n1r, :d].cpu().numpy())
        hir, hdr, hcr = np.zeros(top_k, dtype=np.uint64), np.zeros(top_k, dtype=np.float32), np.zeros(1, dtype=np.uint32)
        hcall = []
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(n1r):
                capi.lib().vers_ivf_search(index._h, Cr_.c_void_p(qhr[i].ctypes.data), 4 * d, 1, top_k, 0, Cr_.c_void_p(hir.ctypes.data), Cr_.c_void_p(hdr.ctypes.data), Cr_.c_void_p(hcr.ctypes.data))
            hcall.append((time.perf_counter() - t0) / n1r)
        e1r_host = float(np.median(hcall))
        # one more batch, kept for the bitwise comparison with vo_search in the CPU leg
        refo = dict(ids=torch.zeros(B, top_k, dtype=torch.int64, device=dev), dst=torch.zeros(B, top_k, device=dev), cnt=torch.zeros(B, dtype=torch.int32, device=dev))
        index.search_dev(Q[(last % n_batches) * B:].data_ptr(), ld, B, top_k, 0, refo["ids"].data_ptr(), refo["dst"].data_ptr(), refo["cnt"].data_ptr(), st)
        index.poll(st)
        extra["reference_mode"] = {"workload": "search_approximate exactly as the reference walks it (nprobe = 0: nearest list + spill), same index / batches",
                                   "batch_ms_per_step": round(tr / args.steps * 1e3, 4), "batch_queries_per_sec": round(args.steps * B / tr, 1),
                                   "list_scan_ms": round(float(np.mean(msr)), 4) if len(msr) else None, "steps": args.steps,
                                   "single_query_end_to_end_us": round(e1r * 1e6, 1), "single_query_host_call_us": round(e1r_host * 1e6, 1)}
        log(f"[bench] reference mode (nprobe=0): {extra['reference_mode']['batch_queries_per_sec']} q/s in batches of {B}, {e1r * 1e6:.1f} us per single query resident, {e1r_host * 1e6:.1f} us per host-pointer call")
    if rank == 0 and not multi and not args.no_extra:
        # (a) single query (B = 1): the list-scan kernel alone (HIP events around its launch) over distinct queries, priced
        # on the bytes of the lists each query actually probed; and the pipelined end-to-end time per query
        nq1 = min(64, B)
        import ctypes as C_
        n_e2e = 200
        nh = min(n_e2e, int(Q.shape[0]))
        qh1 = np.ascontiguousarray(Q[:nh, :d].cpu().numpy())
        hi_, hd_, hc_ = np.zeros(top_k, dtype=np.uint64), np.zeros(top_k, dtype=np.float32), np.zeros(1, dtype=np.uint32)
        lib_ = capi.lib()
        hp = [C_.c_void_p(qh1[i].ctypes.data) for i in range(nh)]
        pi_, pd_, pc_ = C_.c_void_p(hi_.ctypes.data), C_.c_void_p(hd_.ctypes.data), C_.c_void_p(hc_.ctypes.data)

        def single_leg(on_shadow):
            """list scan alone (HIP events), pipelined resident calls, host-pointer calls -- with the single query's scan on the
            fp16 shadow (round 5: scan1h_kernel + the exact finish) or on the f32 rows (rounds 1-4: scan1_kernel + merge)"""
            capi.set_option("single_shadow", 1 if on_shadow else 0)
            row_bytes = (d * 2 + 4) if on_shadow else d * 4   # (shadow row + its |x|^2 | f32 row)
            ms1, by1 = [], []
            capi.set_option("scan_events", 1)   # (single-query calls are not bracketed by event records unless asked: 5.5-6 us per call)
            for i in range(nq1 + 4):
                index.search_dev(Q[i:].data_ptr(), ld, 1, top_k, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
                l1 = index.last_scan()
                if i >= 4:
                    ms1.append(l1["ms"]); by1.append(l1["union_rows"] * row_bytes)
            index.poll(st)
            capi.set_option("scan_events", 2)
            e2e_reps = []
            for rep in range(3):   # (three stretches of 200 queries: the median is reported, the spread kept beside it -- one stretch is 18 ms
                torch.cuda.synchronize(); t0 = time.perf_counter()   # of wall clock and a single hiccup of the host moved it by 10 %)
                for i in range(n_e2e):
                    index.search_dev(Q[(i % (n_batches * B)):].data_ptr(), ld, 1, top_k, nprobe, ids.data_ptr(), dst.data_ptr(), cnt.data_ptr(), st)
                torch.cuda.synchronize(); e2e_reps.append((time.perf_counter() - t0) / n_e2e)
            index.poll(st)
            # the drop-in call itself: vers_ivf_search with HOST pointers, one query per call -- what Index::search_approximate
            # (ivfflat.rs:153) is behind the Rust shim: query in through the pinned block, the launches, result out, the host spinning on the status word
            host_reps = []
            for rep in range(3):
                t0 = time.perf_counter()
                for i in range(nh):
                    lib_.vers_ivf_search(index._h, hp[i], 4 * d, 1, top_k, nprobe, pi_, pd_, pc_)
                host_reps.append((time.perf_counter() - t0) / nh)
            gbs = float(np.sum(by1)) / (float(np.sum(ms1)) * 1e-3) / 1e9
            return {"list_scan_us": round(float(np.mean(ms1)) * 1e3, 1), "probed_list_bytes": int(np.mean(by1)), "achieved_GBs": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "end_to_end_us": round(float(np.median(e2e_reps)) * 1e6, 1),
                    "end_to_end_qps": round(1.0 / float(np.median(e2e_reps)), 1), "end_to_end_us_stretches": [round(x * 1e6, 1) for x in e2e_reps],
                    "host_call_us": round(float(np.median(host_reps)) * 1e6, 1), "host_call_us_stretches": [round(x * 1e6, 1) for x in host_reps]}
        on_shadow = bool(index.shadow_state()["active"]) and nprobe != 0 and top_k + 16 <= 64 and capi.env_option("single_shadow", 1) != 0
        try:
            f32_leg = single_leg(False)
            sq = single_leg(True) if on_shadow else dict(f32_leg)
        finally:
            capi.set_option("single_shadow", 1)
        sq["kernel"] = ("scan1h_kernel (one query on the fp16 shadow: a 64-row tile per wave, VALU dot products) + ivf_rescore_kernel<16> (exact finish)" if on_shadow else
                        "scan1_kernel<0> (ordered f32 chains, one query, one 64-row tile per wave)")
        sq["queries"] = nq1
        sq["host_call"] = "vers_ivf_search (host pointers, one query per call, synchronous): the trait call behind the Rust shim, ctypes overhead included"
        if on_shadow:
            sq["f32_rows_path"] = dict(f32_leg, kernel="scan1_kernel<0> + ivf_merge_kernel (rounds 1-4; vers_set_option('single_shadow', 0)), same process")
        extra["single_query"] = sq
        log(f"[bench] single query: list scan {sq['list_scan_us']} us for {sq['probed_list_bytes'] / 1e6:.0f} MB = {sq['achieved_GBs']:.0f} GB/s "
            f"({sq['frac']:.2f} of peak); end to end {sq['end_to_end_us']} us per query resident, {sq['host_call_us']} us per host-pointer call"
            + (f"  [f32 rows: scan {f32_leg['list_scan_us']} us, {f32_leg['end_to_end_us']} / {f32_leg['host_call_us']} us]" if on_shadow else ""))
        # (a2) between batch 1 and the headline's batch: queries/s, us per batch and which list scan ran (the reference's interface is
        # per query, ivfflat.rs:153: small batches are the realistic serving shape).  One batch per size is kept for the CPU leg.
        sweep, sweep_keep = {}, {}
        for bsz in (1, 8, 32, 128, 512, 1024):
            if bsz > B:
                continue
            def sw_step(i, bsz=bsz):
                o = outs[i % S]
                index.search_dev(Q[(i * bsz) % (n_batches * B - bsz + 1):].data_ptr(), ld, bsz, top_k, nprobe, o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(),
                                 streams[i % S].cuda_stream)
            for i in range(4):
                sw_step(i)
            torch.cuda.synchronize(); pb0 = index.prescan_stats()["batches"]; nst = 40 if bsz <= 128 else 12; t0 = time.perf_counter()
            for i in range(nst):
                sw_step(4 + i)
            torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / nst
            for x in streams:
                index.poll(x.cuda_stream)
            on_mc = index.prescan_stats()["batches"] - pb0 == nst
            sweep[str(bsz)] = {"us_per_batch": round(dt_ * 1e6, 1), "queries_per_sec": round(bsz / dt_, 1),
                               "list_scan": "scan1h_kernel (fp16 shadow, one query) + exact finish" if on_mc and bsz == 1 else
                                            "prescan_kernel_g (matrix cores, fp16 shadow) + exact finish" if on_mc else
                                            ("scan1_kernel (single-query item records)" if bsz == 1 else "scan_kernel (ordered chains)")}
            kq = Q[(last % n_batches) * B:(last % n_batches) * B + bsz]
            ki = torch.zeros(bsz, top_k, dtype=torch.int64, device=dev); kd = torch.zeros(bsz, top_k, device=dev); kc = torch.zeros(bsz, dtype=torch.int32, device=dev)
            index.search_dev(kq.data_ptr(), ld, bsz, top_k, nprobe, ki.data_ptr(), kd.data_ptr(), kc.data_ptr(), st)
            index.poll(st)
            sweep_keep[bsz] = (ki.cpu().numpy().astype(np.uint64), kd.cpu().numpy(), kc.cpu().numpy())
        extra["batch_sweep"] = {"workload": f"the headline's index and queries at other batch sizes, {S} batches in flight, nprobe={nprobe} top_k={top_k}", "by_batch": sweep}
        log("[bench] batch sweep: " + ", ".join(f"{k_}: {v_['queries_per_sec'] / 1e3:.1f} k q/s ({v_['us_per_batch']} us)" for k_, v_ in sweep.items()))
        # (a2b) the EDGES of the fast domain on the headline's index (the reference has no caps: ivfflat.rs:153): results wider than one key per
        # lane (top_k 64 .. 200: candidate lists four keys per lane wide since round 6 -- the ordered chains, 64 ranks per pass, before), more
        # probes than a key per lane (nprobe 128: ranked and scanned on the matrix cores; 256: ranked exactly, scanned there) and batches of
        # 2 - 3 (below pre_min_batch: consecutive single queries on the shadow).  Correctness of these shapes is tests/test_limits_gpu.py's;
        # here: what they cost.
        edges = {}
        for name, (bsz, tk, npb) in {"top_k_64": (min(B, 256), 64, nprobe), "top_k_100": (min(B, 256), 100, nprobe), "top_k_128": (min(B, 256), 128, nprobe), "top_k_200": (min(B, 256), 200, nprobe), "nprobe_128": (min(B, 256), top_k, min(128, nlist)), "nprobe_256": (min(B, 256), top_k, min(256, nlist)),
                                      "batch_2": (2, top_k, nprobe), "batch_3": (3, top_k, nprobe)}.items():
            if bsz > B or npb < 1:
                continue
            ei = torch.zeros(bsz, tk, dtype=torch.int64, device=dev); ed = torch.zeros(bsz, tk, device=dev); ec = torch.zeros(bsz, dtype=torch.int32, device=dev)
            def ed_step(i, bsz=bsz, tk=tk, npb=npb):
                index.search_dev(Q[(i * bsz) % (n_batches * B - bsz + 1):].data_ptr(), ld, bsz, tk, npb, ei.data_ptr(), ed.data_ptr(), ec.data_ptr(), st)
            for i in range(2):
                ed_step(i)
            torch.cuda.synchronize(); pb0 = index.prescan_stats()["batches"]; nst = 6 if bsz > 3 else 30; t0 = time.perf_counter()
            for i in range(nst):
                ed_step(2 + i)
            torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / nst
            index.poll(st)
            edges[name] = {"batch": bsz, "top_k": tk, "nprobe": npb, "us_per_batch": round(dt_ * 1e6, 1), "queries_per_sec": round(bsz / dt_, 1),
                           "list_scan": ("matrix cores + exact finish" if index.prescan_stats()["batches"] - pb0 == nst else
                                         "consecutive single queries on the fp16 shadow (scan1h_kernel) + exact finish" if index.prescan_stats()["batches"] - pb0 == nst * bsz else "ordered chains")}
        extra["domain_edges"] = {"workload": "the headline's index, one batch in flight; shapes at the edges of the matrix-core scan's domain (one key per lane up to top_k = 48, four per lane up to 200; nprobe <= 1024; batch >= 4)", "by_shape": edges}
        log("[bench] domain edges: " + ", ".join(f"{k_}: {v_['queries_per_sec'] / 1e3:.1f} k q/s ({v_['list_scan']})" for k_, v_ in edges.items()))
        # (a3) d = 1536 -- a dimension the reference's own bindings instantiate (vers-py/src/lib.rs:26-65).  A 32-query block with both
        # halves of the query's fp16 hi + lo split does not fit LDS there (196 KB); round 4 ran 16-query blocks (every list probed by
        # more than 16 queries streamed once per extra group: streamed / union rows 2.15, 0.37 of the HBM roofline); round 5 keeps 32
        # queries per block with the query as fp16 hi ONLY (98 KB) and charges the query's measured fp16 residual to the certificate.
        # The headline's geometry at half the rows: the same nlist / nprobe / batch, i.e. the same queries per list and bytes per list.
        d15 = 1536
        n15 = max(4096, min(5_000_000, n // 2)); nl15 = nlist
        X15 = torch.empty(n15, d15, dtype=torch.float32, device=dev)
        capi.gen_rows_dev(X15.data_ptr(), n15, d15, d15, 1, SEED_X + 0x1536, SEED_C, args.modes_per_list * nl15, float(dg.default_sigma(d15)))
        i15 = IVFFlatIndex(d15, device=dev_index)
        # init draws: one row out of every n15 / nl15, i.e. nl15 DISTINCT modes of the generator (row i belongs to mode i % n_modes).
        # (Round 4 drew mix64(0xB15 + c) % n15: a few draws fell into the same mode, their clusters ran empty, the reference's
        # update_centroids turns an empty cluster into the ZERO vector (ivfflat.rs:62-66) and that centroid -- at distance 1 from every
        # unit vector, nearer than any unrelated mode -- collected 47 k rows and was probed by all 1024 queries of a batch: 32 query
        # groups streaming one list, streamed / union rows 1.5-2.2.  The headline's own draws give lists of 917 .. 4604 rows; so do these.)
        init15 = (np.arange(nl15, dtype=np.uint64) * np.uint64(max(1, n15 // nl15))).astype(np.uint64)
        t0 = time.perf_counter(); i15.build_dev(X15.data_ptr(), n15, nl15, 1, args.kmeans_iters, init15); t_b15 = time.perf_counter() - t0
        del X15
        torch.cuda.empty_cache()
        Q15 = torch.empty(4 * B, d15, dtype=torch.float32, device=dev)
        capi.gen_rows_dev(Q15.data_ptr(), 4 * B, d15, d15, 1, SEED_Q + 0x1536, SEED_C, args.modes_per_list * nl15, float(dg.default_sigma(d15)))
        np15 = min(nprobe, nl15)
        def s15(i, S_=S):
            o = outs[i % S_]
            i15.search_dev(Q15[(i % 4) * B:].data_ptr(), d15, B, top_k, np15, o["ids"].data_ptr(), o["dst"].data_ptr(), o["cnt"].data_ptr(), streams[i % S_].cuda_stream)
        for i in range(3):
            s15(i)
        torch.cuda.synchronize(); pb0 = i15.prescan_stats()["batches"]; t0 = time.perf_counter()
        for i in range(10):
            s15(3 + i)
        torch.cuda.synchronize(); t15 = (time.perf_counter() - t0) / 10
        on_mc15 = i15.prescan_stats()["batches"] - pb0 == 10
        for x in streams:
            i15.poll(x.cuda_stream)
        i15.scan_times(reset=True)
        for i in range(6):
            s15(i, 1)
        torch.cuda.synchronize(); i15.poll(streams[0].cuda_stream)
        l15 = i15.last_scan(); ms15 = i15.scan_times(reset=True)[-4:]
        l15_lens = i15.list_lengths()
        by15 = l15["union_rows"] * (d15 * 2 + 4) + nl15 * d15 * 4
        m15 = float(np.mean(ms15)) if len(ms15) else float("nan")
        e15 = {"workload": f"IVFFlat N={n15} d={d15} nlist={nl15} nprobe={np15} batch={B} top_k={top_k}, {S} batches in flight", "ms_per_step": round(t15 * 1e3, 4),
               "queries_per_sec": round(B / t15, 1), "list_scan": ("prescan_kernel_g<true, 32, IvfSrc<32>, LO = false> (32-query blocks, query block as fp16 hi only, fp16 shadow rows) + exact finish" if i15.shadow_state()["active"] else
                             "prescan_kernel_g<false, 16> (narrow 16-query blocks, f32 rows) + exact finish") if on_mc15 else "scan_kernel (ordered chains)",
               "rescanned_queries": int(i15.prescan_stats()["fallback_queries"]),
               "list_scan_ms_one_batch_in_flight": round(m15, 4), "algorithmic_bytes_per_launch": int(by15), "achieved_GBs": round(by15 / (m15 * 1e-3) / 1e9, 1),
               "frac": round(by15 / (m15 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "streamed_over_union_rows": round(l15["streamed_rows"] / max(1, l15["union_rows"]), 3),
               "build_index_s": round(t_b15, 2), "list_len_min_mean_max": [int(l15_lens.min()), int(l15_lens.mean()), int(l15_lens.max())]}
        if not args.no_cpu:   # 32 queries spread over the last batch against the CPU restatement, bit for bit
            from oracle import c_oracle as co15
            c15 = np.ascontiguousarray(i15.get_centroids()); q15 = Q15[(5 % 4) * B:(5 % 4) * B + B].cpu().numpy()
            g_i, g_d, g_c = outs[0]["ids"].cpu().numpy().astype(np.uint64), outs[0]["dst"].cpu().numpy(), outs[0]["cnt"].cpu().numpy()
            ok15 = True
            pick15 = sorted(set(np.linspace(0, B - 1, min(B, 32)).astype(int).tolist()))
            e15["queries_compared_bitwise"] = len(pick15)
            for qi in pick15:
                ranked, _ = co15.search_exhaustive(c15, q15[qi], nl15)
                lists15 = [i15.get_list(int(c_)) for c_ in ranked[:np15]]
                vals = np.concatenate([r_[0] for r_ in lists15]); vid = np.concatenate([r_[1] for r_ in lists15])
                ids_l = [[] for _ in range(nl15)]
                off_ = 0
                for c_, r_ in zip(ranked[:np15], lists15):
                    ids_l[int(c_)] = list(range(off_, off_ + len(r_[1]))); off_ += len(r_[1])
                oi_, od_ = co15.search_nprobe(vals, c15, ids_l, q15[qi], top_k, np15)
                ok15 &= bool(g_c[qi] == len(oi_) and np.array_equal(vid[oi_.astype(np.int64)], g_i[qi, :len(oi_)]) and np.array_equal(od_.view(np.uint32), g_d[qi, :len(od_)].view(np.uint32)))
            e15["gpu_matches_cpu_bitwise"] = ok15
        extra["d1536"] = e15
        log(f"[bench] d = 1536 (N={n15}, nlist={nl15}): {e15['queries_per_sec'] / 1e3:.1f} k q/s, list scan {m15:.3f} ms = {e15['frac']} of peak ({e15['list_scan']})")
        i15.close(); del Q15
        torch.cuda.empty_cache()
        # (b) BASELINE.json cfg2: brute-force scan N = 1M, d = 128, one query; FOUR corpora in rotation (2 GB > the 256 MiB
        # Infinity Cache), kernel time from HIP events
        n2, d2, rot = 1_000_000, 128, 4
        flats = []
        for r in range(rot):
            Xf = torch.empty(n2, d2, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xf.data_ptr(), n2, d2, d2, 0, SEED_X + 0x100 + r)
            fc = capi.FlatCorpus(d2, device=dev_index); fc.upload_dev(Xf.data_ptr(), n2, d2); flats.append(fc)
            del Xf
        Qf = torch.empty(64, d2, dtype=torch.float32, device=dev)
        capi.gen_rows_dev(Qf.data_ptr(), 64, d2, d2, 0, SEED_Q + 0x100)
        fi = torch.zeros(1, top_k, dtype=torch.int64, device=dev); fd = torch.zeros(1, top_k, device=dev); fcn = torch.zeros(1, dtype=torch.int32, device=dev)
        def flat_leg(on_shadow):
            """scan kernel alone (HIP events) per metric and the whole call per query (scan + finish), one query at a time"""
            capi.set_option("single_shadow", 1 if on_shadow else 0)
            ms_ = {0: [], 1: []}
            for metric in (0, 1):
                for i in range(8 + 64):
                    fc = flats[i % rot]
                    fc.search_dev(Qf[i % 64:].data_ptr(), d2, 1, top_k, metric, fi.data_ptr(), fd.data_ptr(), fcn.data_ptr(), st)
                    fc.poll(st)
                    if i >= 8:
                        ms_[metric].append(fc.last_scan_ms())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(200):
                flats[i % rot].search_dev(Qf[i % 64:].data_ptr(), d2, 1, top_k, 0, fi.data_ptr(), fd.data_ptr(), fcn.data_ptr(), st)
            torch.cuda.synchronize(); call = (time.perf_counter() - t0) / 200
            flats[0].poll(st)
            by = n2 * (d2 * 2 + 4) if on_shadow else n2 * d2 * 4   # (shadow row + its |x|^2 | f32 row)
            leg = {"algorithmic_bytes": by, "call_us": round(call * 1e6, 1), "queries_per_sec": round(1.0 / call, 1)}
            for metric, name in ((0, "l2sq"), (1, "cosdist")):
                m_ = float(np.mean(ms_[metric]))
                leg[name] = {"scan_us": round(m_ * 1e3, 1), "achieved_GBs": round(by / (m_ * 1e-3) / 1e9, 1), "frac": round(by / (m_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            return leg
        flat_shadow = capi.env_option("shadow", 1) != 0 and capi.env_option("single_shadow", 1) != 0 and top_k + 16 <= 64
        try:
            f32_leg = flat_leg(False)
            sh_leg = flat_leg(True) if flat_shadow else None
        finally:
            capi.set_option("single_shadow", 1)
        for fc in flats:
            fc.close()
        fb = n2 * d2 * 4
        # (the f32 ordered-chain scan keeps the keys it always had -- SURVEY 8d prices cfg2 on N d 4 bytes; the shadow path of round 5 beside it)
        extra["flat_cfg2"] = {"workload": f"brute-force scan N={n2} d={d2} f32, one query, {rot} corpora in rotation", "algorithmic_bytes": fb,
                              "kernel": "scan_kernel<1, 0, FlatSrc<1>> (ordered f32 chains) + flat_merge_kernel; vers_set_option('single_shadow', 0)" if flat_shadow else "scan_kernel<1, 0, FlatSrc<1>> (ordered f32 chains) + flat_merge_kernel",
                              "l2sq": f32_leg["l2sq"], "cosdist": f32_leg["cosdist"], "call_us": f32_leg["call_us"], "queries_per_sec": f32_leg["queries_per_sec"]}
        if sh_leg:
            extra["flat_cfg2"]["on_the_shadow"] = dict(sh_leg, kernel="flat1h_kernel (the corpus' fp16 shadow, a persistent grid of 64-row tiles per wave, VALU dot products) + ivf_rescore_kernel<16> (exact finish) + fallback_kernel: the default since round 5, same results")
        log(f"[bench] cfg2 flat scan (f32 rows): {f32_leg['l2sq']['scan_us']} us = {f32_leg['l2sq']['frac']} of peak (L2), {f32_leg['cosdist']['scan_us']} us (1 - dot), {f32_leg['call_us']} us per call"
            + (f"; on the shadow: scan {sh_leg['l2sq']['scan_us']} us = {sh_leg['l2sq']['frac']} of peak on its {sh_leg['algorithmic_bytes'] / 1e6:.0f} MB, {sh_leg['call_us']} us per call" if sh_leg else ""))
        # (c) recall on the worst case: Dist-U (uniform on the sphere -- nothing for the lists to cluster on), same N / nlist / nprobe
        if not args.no_recall:
            Xu = torch.empty(n, ld, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xu.data_ptr(), n, d, ld, 0, SEED_X + 7)
            iu = IVFFlatIndex(d, device=dev_index)
            iu.build_dev(Xu.data_ptr(), n, nlist, 1, args.kmeans_iters, init)
            del Xu
            torch.cuda.empty_cache()
            nq_u = min(B, 256)
            Qu = torch.empty(nq_u, ld, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Qu.data_ptr(), nq_u, d, ld, 0, SEED_Q + 7)
            ui = torch.zeros(nq_u, top_k, dtype=torch.int64, device=dev); ud = torch.zeros(nq_u, top_k, device=dev); uc = torch.zeros(nq_u, dtype=torch.int32, device=dev)
            ei = torch.zeros(nq_u, top_k, dtype=torch.int64, device=dev); ed = torch.zeros(nq_u, top_k, device=dev); ec = torch.zeros(nq_u, dtype=torch.int32, device=dev)
            iu.search_dev(Qu.data_ptr(), ld, nq_u, top_k, nprobe, ui.data_ptr(), ud.data_ptr(), uc.data_ptr(), st)
            iu.search_exhaustive_dev(Qu.data_ptr(), ld, nq_u, top_k, 0, ei.data_ptr(), ed.data_ptr(), ec.data_ptr(), st)
            iu.poll(st)
            a_, e_ = ui.cpu().numpy(), ei.cpu().numpy()
            ru = sum(len(set(a_[q].tolist()) & set(e_[q].tolist())) for q in range(nq_u)) / float(nq_u * top_k)
            extra["recall_at_10_dist_u"] = {"value": round(ru, 4), "queries": nq_u, "note": "uniform sphere: the worst case for an inverted file "
                                            f"(nprobe/nlist = {nprobe}/{nlist} of the lists); the headline corpus is Dist-C"}
            log(f"[bench] recall@{top_k} on Dist-U (uniform sphere), same N / nlist / nprobe: {ru:.4f} over {nq_u} queries")
            iu.close()
            torch.cuda.empty_cache()
            # (d) between the two: Dist-C with TWICE the noise (its norm about equals the centre's: the modes overlap), 10 k-means
            # iterations, recall@10 against the exact scan as a function of nprobe
            Xm = torch.empty(n, ld, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xm.data_ptr(), n, d, ld, 1, SEED_X + 9, SEED_C, n_modes, 2.0 * sigma)
            im = IVFFlatIndex(d, device=dev_index)
            t0 = time.perf_counter()
            im.build_dev(Xm.data_ptr(), n, nlist, 1, 10, init)
            t_bm = time.perf_counter() - t0
            del Xm
            torch.cuda.empty_cache()
            nq_m = min(B, 256)
            Qm = torch.empty(nq_m, ld, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Qm.data_ptr(), nq_m, d, ld, 1, SEED_Q + 9, SEED_C, n_modes, 2.0 * sigma)
            mi = torch.zeros(nq_m, top_k, dtype=torch.int64, device=dev); md = torch.zeros(nq_m, top_k, device=dev); mc = torch.zeros(nq_m, dtype=torch.int32, device=dev)
            xi = torch.zeros(nq_m, top_k, dtype=torch.int64, device=dev); xd = torch.zeros(nq_m, top_k, device=dev); xc = torch.zeros(nq_m, dtype=torch.int32, device=dev)
            im.search_exhaustive_dev(Qm.data_ptr(), ld, nq_m, top_k, 0, xi.data_ptr(), xd.data_ptr(), xc.data_ptr(), st)
            im.poll(st)
            e_ = xi.cpu().numpy()
            table = {}
            for npb in (1, 8, 32, 128):
                im.search_dev(Qm.data_ptr(), ld, nq_m, top_k, min(npb, nlist), mi.data_ptr(), md.data_ptr(), mc.data_ptr(), st)
                im.poll(st)
                a_ = mi.cpu().numpy(); c_ = mc.cpu().numpy()
                table[str(npb)] = round(sum(len(set(a_[q, :c_[q]].tolist()) & set(e_[q].tolist())) for q in range(nq_m)) / float(nq_m * top_k), 4)
            extra["recall_vs_nprobe_noisy_dist_c"] = {"recall_at_10": table, "queries": nq_m, "kmeans_iterations": int(im.iterations[0]), "build_index_s": round(t_bm, 2),
                                                      "distribution": f"Dist-C, {n_modes} modes, sigma x 2 (noise norm ~ centre norm), same N / d / nlist"}
            log(f"[bench] recall@{top_k} vs nprobe on the noisy Dist-C (sigma x 2, {int(im.iterations[0])} k-means iterations): {table}")
            im.close()
            torch.cuda.empty_cache()

    # ---- CPU baseline: the C restatement of the reference path (oracle/vers_oracle.c) on the host cores ----------
    # Per sampled query the lists the reference would touch are read back from HBM into a sub-index with all nlist
    # centroids (the other lists stay empty) -- not timed -- and ONE call of vo_search_nprobe / vo_search on it is timed:
    # every centroid distance + stable sort, the per-candidate row gather through the id lists (ivfflat.rs:172-175),
    # every row distance + stable sort.  First on one thread (the reference's search_approximate is serial), then the same
    # call for independent queries on all host cores ("embarrassingly parallel over queries").
    cpu, cpu_all, cpu_km = None, None, None
    if rank == 0 and not multi and not args.no_cpu:
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        from oracle import c_oracle as co
        cent = np.ascontiguousarray(index.get_centroids())
        qh = Q[(last % n_batches) * B:(last % n_batches) * B + B, :d].cpu().numpy()
        fp_, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64)

        def sub_index(q, nprobe=nprobe):
            ranked, _ = co.search_exhaustive(cent, q, nlist)
            lists, rem = [], top_k
            for c in ranked[:nprobe] if nprobe else ranked:
                rows, rid = index.get_list(int(c))
                lists.append((int(c), rows, rid))
                if not nprobe:
                    rem -= min(rem, len(rid))
                    if rem == 0:
                        break
            off = np.zeros(nlist + 1, dtype=np.uint64)
            for c, rows, rid in lists:
                off[c + 1] = len(rid)
            off = np.cumsum(off).astype(np.uint64)
            vals = np.zeros((max(1, int(off[-1])), d), dtype=np.float32); vid = np.zeros(max(1, int(off[-1])), dtype=np.uint64)
            for c, rows, rid in lists:
                vals[int(off[c]):int(off[c + 1])] = rows; vid[int(off[c]):int(off[c + 1])] = rid
            return vals, vid, off, np.arange(max(1, int(off[-1])), dtype=np.uint64)

        def run_one(q, sub, nprobe=nprobe):
            vals, vid, off, loc = sub
            oi = np.zeros(max(1, top_k), dtype=np.uint64); od = np.zeros(max(1, top_k), dtype=np.float32)
            t0 = time.perf_counter()
            if nprobe:
                m = co.lib().vo_search_nprobe_m(vals.ctypes.data_as(fp_), cent.ctypes.data_as(fp_), nlist, d, off.ctypes.data_as(u64p),
                                                loc.ctypes.data_as(u64p), q.ctypes.data_as(fp_), top_k, nprobe, oi.ctypes.data_as(u64p),
                                                od.ctypes.data_as(fp_), 0)
            else:
                m = co.lib().vo_search_m(vals.ctypes.data_as(fp_), cent.ctypes.data_as(fp_), nlist, d, off.ctypes.data_as(u64p),
                                         loc.ctypes.data_as(u64p), q.ctypes.data_as(fp_), top_k, oi.ctypes.data_as(u64p), od.ctypes.data_as(fp_), 0)
            dt = time.perf_counter() - t0
            assert m >= 0, f"oracle status {m}"
            return dt, vid[oi[:m].astype(np.int64)], od[:m]

        t_cpu, n_cpu, mismatches = 0.0, 0, 0
        while t_cpu < args.cpu_seconds and n_cpu < B:
            q = np.ascontiguousarray(qh[n_cpu])
            dt, got_ids, dd = run_one(q, sub_index(q))
            t_cpu += dt
            ok = (np.array_equal(got_ids, ids_h[n_cpu, :len(got_ids)]) and cnt_h[n_cpu] == len(got_ids)
                  and np.array_equal(dd.view(np.uint32), dst_h[n_cpu, :len(dd)].view(np.uint32)))
            mismatches += 0 if ok else 1
            n_cpu += 1
        mode = f"nprobe={nprobe}" if nprobe else "reference mode (nearest list + spill)"
        cpu = {"value": round(n_cpu / t_cpu, 3), "unit": "queries/sec", "cores": 1, "kind": "port",
               "sample": f"{n_cpu} queries of the last timed batch, {mode}: one vo_search{'_nprobe' if nprobe else ''} call per query on a sub-index read "
                         f"back from HBM (all {nlist} centroids, the lists the query touches; incl. centroid ranking, per-candidate row gather, "
                         f"stable sorts); single thread like the reference's serial search_approximate",
               "gpu_matches_cpu_bitwise": mismatches == 0, "mismatching_queries": mismatches}
        log(f"[bench] cpu baseline {cpu['value']} q/s on 1 core over {n_cpu} queries; GPU==CPU bitwise: {mismatches == 0}")
        # all host cores, one query per thread (the C call releases the GIL)
        cores = os.cpu_count() or 1
        model = ""
        try:
            model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except (OSError, StopIteration):
            pass
        per_q = t_cpu / max(1, n_cpu)
        n_par = int(max(min(cores, B), min(4 * cores, (args.cpu_seconds * cores) / max(per_q, 1e-6))))
        n_par = min(n_par, B)
        if n_par > 0:
            pick = [(n_cpu + i) % B for i in range(n_par)]  # further queries of the batch (wrapping around on small batches)
            qs = [np.ascontiguousarray(qh[i]) for i in pick]
            subs = [sub_index(q) for q in qs]                                   # (not timed)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                res = list(ex.map(lambda a: run_one(*a), zip(qs, subs)))
            wall = time.perf_counter() - t0
            bad = sum(0 if (np.array_equal(r[1], ids_h[pick[i], :len(r[1])]) and
                            np.array_equal(r[2].view(np.uint32), dst_h[pick[i], :len(r[2])].view(np.uint32))) else 1 for i, r in enumerate(res))
            cpu_all = {"value": round(n_par / wall, 3), "unit": "queries/sec", "cores": cores, "kind": "port", "cpu_model": model,
                       "sample": f"{n_par} further queries of the same batch, one per thread on {cores} threads (independent searches; the "
                                 f"reference itself searches serially)", "gpu_matches_cpu_bitwise": bad == 0}
            log(f"[bench] cpu baseline on all {cores} host threads ({model}): {cpu_all['value']} q/s over {n_par} queries; GPU==CPU bitwise: {bad == 0}")

        # ---- the batch sweep's kept batches against vo_search_nprobe, two queries per size, bit for bit ------------------------------
        if "batch_sweep" in extra:
            for bsz, (ki_, kd_, kc_) in sweep_keep.items():
                okb = True
                for qi in sorted(set(np.linspace(0, bsz - 1, min(bsz, 8)).astype(int).tolist())):   # (up to 8 queries spread over the kept batch)
                    q = np.ascontiguousarray(qh[qi])
                    _, gi_, gd_ = run_one(q, sub_index(q))
                    okb &= bool(kc_[qi] == len(gi_) and np.array_equal(gi_, ki_[qi, :len(gi_)]) and np.array_equal(gd_.view(np.uint32), kd_[qi, :len(gd_)].view(np.uint32)))
                extra["batch_sweep"]["by_batch"][str(bsz)]["gpu_matches_cpu_bitwise"] = okb
            log(f"[bench] batch sweep: GPU == CPU bitwise at every size: {all(v_['gpu_matches_cpu_bitwise'] for v_ in extra['batch_sweep']['by_batch'].values())}")

        # ---- the reference's own mode: GPU batch of extra.reference_mode against vo_search, bit for bit ----------------------
        if "reference_mode" in extra:
            ri, rd, rc_ = refo["ids"].cpu().numpy().astype(np.uint64), refo["dst"].cpu().numpy(), refo["cnt"].cpu().numpy()
            t_ref, n_ref, bad_ref = 0.0, 0, 0
            while t_ref < max(2.0, args.cpu_seconds / 4) and n_ref < min(B, 256):
                q = np.ascontiguousarray(qh[n_ref])
                dt, gi, gd = run_one(q, sub_index(q, 0), 0)
                t_ref += dt
                ok = len(gi) == rc_[n_ref] and np.array_equal(gi, ri[n_ref, :len(gi)]) and np.array_equal(gd.view(np.uint32), rd[n_ref, :len(gd)].view(np.uint32))
                bad_ref += 0 if ok else 1
                n_ref += 1
            extra["reference_mode"].update({"cpu_queries_per_sec_1_core": round(n_ref / t_ref, 2), "cpu_sample": f"{n_ref} queries, one vo_search call each",
                                            "gpu_matches_cpu_bitwise": bad_ref == 0})
            log(f"[bench] reference mode: CPU restatement {n_ref / t_ref:.1f} q/s on 1 core over {n_ref} queries; GPU==CPU bitwise: {bad_ref == 0}")

        # ---- cfg2 on the host: utils::search_exhaustive (utils.rs:68-82), N = 1M d = 128, against the GPU flat scan ------------
        if not args.no_extra and "flat_cfg2" in extra:
            n2, d2 = 1_000_000, 128
            Xf = torch.empty(n2, d2, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xf.data_ptr(), n2, d2, d2, 0, SEED_X + 0x100)
            fc = capi.FlatCorpus(d2, device=dev_index); fc.upload_dev(Xf.data_ptr(), n2, d2)
            Qf = torch.empty(64, d2, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Qf.data_ptr(), 64, d2, d2, 0, SEED_Q + 0x100)
            Xh, Qh2 = Xf.cpu().numpy(), Qf.cpu().numpy()
            del Xf
            gi2, gd2, _gc2 = fc.search(Qh2, top_k)
            fc.close()
            t2, n2q, bad2 = 0.0, 0, 0
            while t2 < max(2.0, args.cpu_seconds / 4) and n2q < 64:
                t0 = time.perf_counter(); oi, od = co.search_exhaustive(Xh, Qh2[n2q], top_k); t2 += time.perf_counter() - t0
                bad2 += 0 if (np.array_equal(oi, gi2[n2q]) and np.array_equal(od.view(np.uint32), gd2[n2q].view(np.uint32))) else 1
                n2q += 1
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=min(cores, 64)) as ex:
                list(ex.map(lambda i: co.search_exhaustive(Xh, Qh2[i], top_k), range(64)))
            w2 = time.perf_counter() - t0
            extra["flat_cfg2"]["cpu_baseline"] = {"value": round(n2q / t2, 3), "unit": "queries/sec", "cores": 1, "kind": "port", "us_per_query": round(t2 / n2q * 1e6, 0),
                                                  "sample": f"{n2q} queries, one vo_search_exhaustive call each over the same {n2} x {d2} corpus (stable sort of all N distances)",
                                                  "all_cores": {"value": round(64 / w2, 2), "cores": min(cores, 64), "sample": "64 queries, one per thread"},
                                                  "gpu_matches_cpu_bitwise": bad2 == 0}
            log(f"[bench] cfg2 on the host: {n2q / t2:.2f} q/s on 1 core, {64 / w2:.1f} q/s on {min(cores, 64)} threads; GPU==CPU bitwise: {bad2 == 0}")
            del Xh

        # ---- k-means on the host: assign_to_clusters on all cores (= rayon par_iter, ivfflat.rs:31),
        # update_centroids + calculate_kmeans_cost serial like the reference; a bounded sample of N = 1M, k = nlist, same d
        n_km = min(n, 1_000_000)
        Xk = torch.empty(n_km, ld, dtype=torch.float32, device=dev)
        capi.gen_rows_dev(Xk.data_ptr(), n_km, d, ld, 1, SEED_X, SEED_C, n_modes, sigma)
        Xkh = np.ascontiguousarray(Xk[:, :d].cpu().numpy())
        del Xk
        chunk = 4
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:   # first wave: one small task per thread -> the rate
            first = list(ex.map(lambda i: co.assign_to_clusters(Xkh[i * chunk:(i + 1) * chunk], cent), range(cores)))
        rate = cores * chunk / (time.perf_counter() - t0)
        n_as = int(min(n_km, max(cores * chunk, 0.3 * rate * args.cpu_seconds)))  # (the first wave runs hot in cache: the long run is ~3x slower per point)
        per = max(1, n_as // (cores * 4))
        n_as = per * (n_as // per)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            parts = list(ex.map(lambda i: co.assign_to_clusters(Xkh[i * per:(i + 1) * per], cent), range(n_as // per)))
        t_as = time.perf_counter() - t0
        a_cpu = np.concatenate(parts)
        a_gpu = capi.kmeans_assign(Xkh[:n_as], cent, device=dev_index)
        n_uc = min(n_km, 200_000)   # serial legs: a chain of d dependent adds per row (~1 us at d = 768)
        a_all = capi.kmeans_assign(Xkh[:n_uc], cent, device=dev_index)
        t0 = time.perf_counter(); c_cpu = co.update_centroids(Xkh[:n_uc], a_all, nlist); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); cost_cpu = co.kmeans_cost(Xkh[:n_uc], cent, a_all); t_co = time.perf_counter() - t0
        c_gpu = capi.kmeans_update(Xkh[:n_uc], a_all, nlist, device=dev_index)
        cost_gpu = capi.kmeans_cost(Xkh[:n_uc], cent, a_all, device=dev_index)
        pass_flop = 2.0 * n_as * nlist * d
        cpu_km = {"assign_to_clusters": {"points": n_as, "k": nlist, "d": d, "seconds": round(t_as, 2), "cores": cores,
                                         "points_per_sec": round(n_as / t_as, 1), "gflops_2nkd": round(pass_flop / t_as / 1e9, 1),
                                         "seconds_per_pass_at_N_1M": round(1e6 / (n_as / t_as), 1),
                                         "seconds_per_pass_extrapolated_cfg5": round(50e6 * 65536 / (n_as * nlist / t_as), 0),
                                         "gpu_matches_cpu": bool(np.array_equal(a_cpu, a_gpu))},
                  "update_centroids": {"points": n_uc, "seconds": round(t_up, 3), "cores": 1, "seconds_per_pass_at_N_1M": round(t_up * 1e6 / n_uc, 2),
                                       "gpu_matches_cpu_bitwise": bool(np.array_equal(c_cpu.view(np.uint32), c_gpu.view(np.uint32)))},
                  "calculate_kmeans_cost": {"points": n_uc, "seconds": round(t_co, 3), "cores": 1, "seconds_per_pass_at_N_1M": round(t_co * 1e6 / n_uc, 2),
                                            "gpu_matches_cpu_bitwise": bool(np.float32(cost_cpu).view(np.uint32) == np.float32(cost_gpu).view(np.uint32))},
                  "kind": "port", "cpu_model": model,
                  "sample": f"rows 0..{n_as} of the bench corpus against the index's {nlist} centroids for assign (all {cores} threads, one slice per task), "
                            f"rows 0..{n_uc} for the serial update / cost; C restatement of ivfflat.rs:29-71,138-149 (vo_assign / vo_update / vo_cost)"}
        log(f"[bench] k-means on the host: assign {n_as} points x {nlist} centroids in {t_as:.1f} s on {cores} threads = {pass_flop / t_as / 1e9:.0f} GFLOP/s "
            f"(GPU == CPU: {cpu_km['assign_to_clusters']['gpu_matches_cpu']}); update {t_up:.2f} s, cost {t_co:.2f} s for {n_uc} points on 1 core")
        del Xkh

    # ---- Index::add (ivfflat.rs:200-213), one vector per call, host pointer: the LAST thing done to the headline's index ----
    if rank == 0 and not multi and not args.no_extra:
        try:
            import ctypes as C_
            na = 96
            Xa = torch.empty(na, d, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xa.data_ptr(), na, d, d, 1, SEED_X + 0xADD, SEED_C, n_modes, sigma)
            xa = np.ascontiguousarray(Xa.cpu().numpy())
            c_, v_ = C_.c_uint64(0), C_.c_uint64(0)
            ta = []
            for i in range(na):
                t0 = time.perf_counter()
                capi.check(capi.lib().vers_ivf_add(index._h, C_.c_void_p(xa[i].ctypes.data), C_.byref(c_), C_.byref(v_)))
                ta.append(time.perf_counter() - t0)
            extra["add"] = {"us_per_vector": round(float(np.median(ta[32:])) * 1e6, 1), "vectors": na - 32, "last_vec_id": int(v_.value),
                            "what": "vers_ivf_add (host pointer): nearest centroid, append to its list (f32 tiles, shadow, row-major copy, |x|^2, tables), consistent on return"}
            log(f"[bench] add: {extra['add']['us_per_vector']} us per vector")
        except Exception as e:
            log(f"[bench] add leg failed: {e!r}")

    # ---- the COMPACT memory layout (vers_set_option("memory", 1): ONE f32 copy of the rows -- tiles + fp16 shadow, no row-major copy; the
    # exact finish gathers its survivors from the tiles): the headline's index rebuilt under it, the same timed steps, bytes per row
    if rank == 0 and not multi and not args.no_extra:
        try:
            index.close()
            torch.cuda.empty_cache()
            capi.set_option("memory", 1)
            Xc = torch.empty(n, ld, dtype=torch.float32, device=dev)
            capi.gen_rows_dev(Xc.data_ptr(), n, d, ld, 1, SEED_X, SEED_C, n_modes, sigma)
            index = IVFFlatIndex(d, device=dev_index)
            m0, _ = capi.mem_stats(reset_peak=True)
            index.build_dev(Xc.data_ptr(), n, nlist, 1, args.kmeans_iters, init)
            del Xc
            torch.cuda.empty_cache()
            mc_now, mc_peak = capi.mem_stats()
            tcs, msc = timed_steps(nprobe)
            tc1, msc1 = timed_steps(nprobe, 1) if S > 1 else (tcs, msc)
            oi_c, od_c, oc_c = torch.zeros(B, top_k, dtype=torch.int64, device=dev), torch.zeros(B, top_k, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
            index.search_dev(Q[(last % n_batches) * B:].data_ptr(), ld, B, top_k, nprobe, oi_c.data_ptr(), od_c.data_ptr(), oc_c.data_ptr(), st)
            index.poll(st)
            same = bool(np.array_equal(oi_c.cpu().numpy().astype(np.uint64), ids_h) and np.array_equal(od_c.cpu().numpy().view(np.uint32), dst_h.view(np.uint32)) and np.array_equal(oc_c.cpu().numpy(), cnt_h))
            lay_c = index.layout_bytes()
            extra["memory_compact"] = {"library_bytes_now": int(mc_now - m0), "library_bytes_peak_during_build": int(mc_peak - m0), "bytes_per_stored_row": round((mc_now - m0) / n, 1),
                                       "over_f32_rows": round((mc_now - m0) / n / (4.0 * d), 3), "rowmajor_copy_bytes": int(lay_c["rowmajor"]),
                                       "whole_step_ms": round(tcs / args.steps * 1e3, 4), "whole_step_queries_per_sec": round(args.steps * B / tcs, 1),
                                       "one_batch_in_flight_whole_step_ms": round(tc1 / args.steps * 1e3, 4), "list_scan_ms": round(float(np.mean(msc1)), 4) if len(msc1) else None,
                                       "same_results_as_the_headline_bitwise": same, "steps": args.steps, "warmup": args.warmup,
                                       "note": "vers_set_option('memory', 1) before build_index: the same corpus, index, batches and timed loop as the headline (after it)"}
            log(f"[bench] compact memory: {extra['memory_compact']['over_f32_rows']} x the f32 rows ({extra['memory_compact']['bytes_per_stored_row']} B per row), "
                f"{extra['memory_compact']['whole_step_ms']} ms per step ({S} in flight), {extra['memory_compact']['one_batch_in_flight_whole_step_ms']} ms one in flight; same results: {same}")
        except Exception as e:
            extra["memory_compact"] = {"failed": f"{type(e).__name__}: {e}"}
            log(f"[bench] compact memory leg FAILED: {e!r}")
        finally:
            capi.set_option("memory", 0)

    # ---- cfg4 / cfg5 at ONE RANK'S NOMINAL SIZE on this GPU (scripts/rank_nominal.py; no 8-GPU node has been available in any round):
    # rank 0 of 8 of IVFFlat N=100M (the corpus streamed through vers_kmeans_assign_dev and vers_ivf_upload_begin / _chunk_dev / _end,
    # 12.5M rows kept; the rank's step, its scan's roofline fraction, memory, GPU == CPU bitwise over the rank's sub-index) and one
    # rank's 6.25M x 768 rows of the k = 65536 k-means through vers_ivf_build_sharded_dev.  After everything else: the headline's index is gone.
    if rank == 0 and not multi and not args.no_extra and not args.no_nominal and d == 768:
        index.close()
        torch.cuda.empty_cache()
        from scripts import rank_nominal as rn
        # (cfg4's nlist is not specified by BASELINE.json: 16384 per SURVEY.md 8d, and the headline's 4096 for comparability)
        for name, fn in (("cfg4_rank", lambda: rn.cfg4_rank(dev_index, log=log, check=0 if args.no_cpu else 32)),
                         ("cfg4_rank_nlist4096", lambda: rn.cfg4_rank(dev_index, nlist=4096, log=log, check=0 if args.no_cpu else 32)),
                         ("cfg5_rank", lambda: rn.cfg5_rank(dev_index, log=log))):
            t0 = time.perf_counter()
            try:
                extra[name] = fn()
                extra[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
            except Exception as e:  # (a leg that fails must not take the headline with it: it is reported as failed)
                extra[name] = {"failed": f"{type(e).__name__}: {e}"}
                log(f"[bench] {name} FAILED: {e}")
            torch.cuda.empty_cache()

    if rank == 0:
        out = {"metric": "queries/sec + recall@10, IVFFlat N=10M d=768", "value": round(qps, 1), "unit": "queries/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "recall_at_10": None if recall is None else round(recall, 4), "self_retrieval_ok": self_ok,
               "config": {"workload": f"IVFFlat search_approximate, {'nprobe extension' if nprobe else 'reference mode (nearest list + spill)'}: N={n} d={d} nlist={nlist} nprobe={nprobe} "
                                      f"batch={B} top_k={top_k}, f32, clustered unit vectors (Dist-C)",
                          "n": n, "d": d, "nlist": nlist, "nprobe": nprobe, "batch": B, "top_k": top_k,
                          "kmeans_iters": int(index.iterations[0]), "parallelism": f"lists sharded over {world} GPU(s)", "batches_in_flight": S,
                          "exchange": exchange_kind, "exchange_tag": exchange_tag},
               "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "cpu_baseline_kmeans": cpu_km, "extra": extra}
        out["row_operand"] = roofline["row_operand"]   # what the DOMINANT KERNEL streams (dtype above = what the path computes and returns)
        if shadow and "list_scan_f32_rows" in extra:
            out["value_f32_rows"] = extra["list_scan_f32_rows"]["whole_step_queries_per_sec"]  # the same steps with VERS_SHADOW=0, timed like the headline, after it
        if shadow:
            out["result_precision"] = ("every returned id, order and distance is the reference's exact f32 result (compared bitwise with the CPU restatement in "
                                        "this run: cpu_baseline.gpu_matches_cpu_bitwise); the dominant kernel PRE-SELECTS candidates on an fp16 copy of the rows and "
                                        "the exact f32 finish + certificate are inside the timed step; the same step on f32 rows: extra.list_scan_f32_rows")
        emit(out, args.extra_file)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
