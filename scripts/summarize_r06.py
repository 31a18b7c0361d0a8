"""Condenses gpurun_out/prof_r06 (scripts/profile_r06.sh) into the text committed under profiles/:
per-kernel durations from the kernel traces (mean / min over WARM dispatches) and per-dispatch means of the PMC
counters, with the HBM bytes derived as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB-like units of 1 KB and
gfx950 reports half the bytes of wide coalesced streaming reads: x2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("vers::", "")
    return name if len(name) < 100 else name[:97] + "..."


def trace(tag, skip_first=0):
    rows = []
    for f in glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg


def pmc(tag):
    cnt = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return cnt


def show_trace(title, tag, warm, top=14):
    agg = trace(tag)
    print(f"== {title}: kernel trace (us); the first {warm} dispatches of every kernel dropped as warm-up ==")
    print(f"{'calls':>6} {'mean_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    rows = []
    for k, v in agg.items():
        w = v[warm:] if len(v) > warm else v
        rows.append((sum(w), len(w), sum(w) / len(w), min(w), max(w), k))
    for tot, n, mean, mn, mx, k in sorted(rows, reverse=True)[:top]:
        print(f"{n:6d} {mean:10.1f} {mn:10.1f} {mx:10.1f}  {short(k)}")
    print()
    return agg


def mean_of(cnt, kernel_sub, counter, warm=0, stop=None):
    for k, cs in cnt.items():
        if kernel_sub in k and counter in cs:
            v = cs[counter][warm:stop] if len(cs[counter]) > warm else cs[counter]
            return sum(v) / len(v), len(v)
    return None, 0


facts = {}
try:
    print(f"commit {open(os.path.join(out, 'commit.txt')).read().strip()}; sha256[:16] of vers_amd/lib/libvers_hip.so as profiled: {open(os.path.join(out, 'lib_sha16.txt')).read().strip()}\n")
except OSError:
    pass
# ---- cfg3 batch 1024 ----
a3 = show_trace("cfg3 batch 1024, ONE batch in flight (bench.py --streams 1 --steps 20 --warmup 5 --no-extra: ONLY the timed configuration's launches)", "cfg3/trace", 5, top=30)
a3x = show_trace("the same run WITH the extra legs (their kernels: f32 coarse contraction, f32-row scan, batch sweep, d = 1536 ...; the headline kernel's rows here mix batch sizes: do not use)", "cfg3x/trace", 0, top=30)
for k_, v_ in a3x.items():   # (the contractions / f32-row scan exist only in the pass with the extras)
    a3.setdefault(k_, v_)
# the production list-scan kernel: <true = fp16 shadow rows (default), <false = f32 rows (VERS_SHADOW=0; also run by bench.py's extra block)
PK = "prescan_kernel_g<true" if any("prescan_kernel_g<true" in k for k in a3) else "prescan_kernel_g<false"
f3, n3 = mean_of(pmc("cfg3/pmc_fetch"), PK, "FETCH_SIZE", 5, 25)   # (the 20 timed launches, as for the durations)
w3, _ = mean_of(pmc("cfg3/pmc_write"), PK, "WRITE_SIZE", 5, 25)
pk = [v for k, v in a3.items() if PK in k]
if pk and f3:
    d = pk[0][5:25]   # the 20 TIMED launches (5 warm-up before them; behind them the run's self-retrieval check -- a batch of 8 -- and its result read-backs)
    print(f"{PK}...>: mean {sum(d)/len(d):.1f} us over the {len(d)} timed dispatches (min {min(d):.1f}); FETCH_SIZE mean {f3:.0f} KB over {n3} "
          f"dispatches -> HBM read bytes per launch = FETCH_SIZE * 1024 * 2 = {f3*1024*2:.4g}; WRITE_SIZE mean {w3 or 0:.0f} KB")
    facts["cfg3"] = {"kernel": short([k for k in a3 if PK in k][0]).split("(")[0].replace("void ", "") + (" (fp16 shadow rows)" if "true" in PK else " (f32 rows)"), "config": {"rows": 10000000, "d": 768, "nlist": 4096, "nprobe": 32, "batch": 1024},
                     "FETCH_SIZE_KB_mean": f3, "WRITE_SIZE_KB_mean": w3, "hbm_read_bytes_per_launch": int(f3 * 1024 * 2),
                     "kernel_mean_us": sum(d) / len(d), "kernel_min_us": min(d),
                     "correction": "gfx950 reports half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section): x2",
                     "algorithmic_bytes_check": "kernel_mean_us is from the CLEAN trace pass (--no-extra): bytes / time <= the 8 TB/s peak by construction",
                     "source": "profiles/r06_summary.txt (rocprofv3 --kernel-trace pass and --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu --no-recall --no-extra`)"}
m = pmc("cfg3/pmc_mfma")
for sub, label, flops in (("dist_gemm_x3_kernel<false", "coarse contraction [1024 x 768].[768 x 4096] as 3 bf16 products (production)", 2 * 1024 * 4096 * 768),
                          ("dist_gemm_kernel<false", "coarse contraction, f32 MFMA kernel (bench.py extra.coarse_gemm_f32)", 2 * 1024 * 4096 * 768),
                          ("prescan_kernel_g<true", "list scan, fp16 shadow rows (production)", None),
                          ("prescan_kernel_g<false", "list scan, f32 rows (bench.py extra.list_scan_f32_rows)", None)):
    busy, _ = mean_of(m, sub, "SQ_VALU_MFMA_BUSY_CYCLES", 2); gui, _ = mean_of(m, sub, "GRBM_GUI_ACTIVE", 2)
    d = [v for k, v in a3.items() if sub in k]
    if busy and gui and d:
        dd = d[0][5:] if len(d[0]) > 5 else d[0][1:] if len(d[0]) > 1 else d[0]
        extra = f"; {flops/(sum(dd)/len(dd))/1e6:.1f} algorithmic TFLOP/s mean, {flops/min(dd)/1e6:.1f} best = {flops/(sum(dd)/len(dd))/1e6/157.3*100:.1f} % / {flops/min(dd)/1e6/157.3*100:.1f} % of the 157.3 TFLOP/s f32 MFMA peak" if flops else ""
        # busy cycles are summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs
        print(f"{label}: SQ_VALU_MFMA_BUSY_CYCLES {busy:.4g} / 1024 SIMDs = {busy/1024:.4g} busy cycles per SIMD; GRBM_GUI_ACTIVE {gui:.4g} / 8 XCDs = "
              f"{gui/8:.4g} kernel cycles -> MFMA-busy {busy/1024/(gui/8)*100:.1f} %; kernel mean {sum(dd)/len(dd):.1f} us (trace){extra}")
sq = pmc("cfg3/pmc_sq")
for sub in ("prescan_kernel_g<true", "prescan_kernel_g<false", "coarse_select_rescore", "ivf_rescore_kernel", "group_scatter_kernel", "dist_gemm_x3_kernel<false", "dist_gemm_x3_kernel<true"):
    warm = 1 if sub == "prescan_kernel_g<false" else 5
    wc, _ = mean_of(sq, sub, "SQ_WAVE_CYCLES", warm); va, _ = mean_of(sq, sub, "SQ_ACTIVE_INST_VALU", warm); wa, _ = mean_of(sq, sub, "SQ_WAIT_ANY", warm)
    if wc:
        print(f"{sub}: VALU-active {100*(va or 0)/wc:.0f} % of wave cycles, waiting (any) {100*(wa or 0)/wc:.0f} %")
print()
# ---- single query ----
a1 = show_trace("cfg3 single query (bench.py --batch 1 --steps 300 --warmup 20)", "b1/trace", 20, top=8)
for nm in ("coarse1_kernel", "scan1h_kernel", "ivf_rescore_kernel<16>", "fallback_kernel", "scan1_kernel", "ivf_merge_kernel"):  # the query's own launches (the table above is led by the build)
    for k, v in a1.items():
        if nm in k and len(v) > 20:
            d = v[20:]
            print(f"  {short(k)[:70]:70s} mean {sum(d)/len(d):6.1f} us (min {min(d):.1f}) over {len(d)} warm dispatches")
s1name = "scan1h_kernel" if any("scan1h_kernel" in k for k in a1) else "scan1_kernel"  # (the fp16 shadow scan of round 6, or the f32 ordered-chain scan)
f1, n1 = mean_of(pmc("b1/pmc_fetch"), s1name, "FETCH_SIZE", 20)
k1 = [v for k, v in a1.items() if s1name in k or (s1name == "scan1_kernel" and "scan_kernel<1, 0, IvfSrc<1>" in k.replace("vers::", ""))]
if k1 and f1:
    d = k1[0][20:]
    by = f1 * 1024 * 2
    print(f"{s1name}: mean {sum(d)/len(d):.1f} us (min {min(d):.1f}) over {len(d)} warm dispatches; FETCH_SIZE mean {f1:.0f} KB -> "
          f"{by/1e6:.0f} MB of HBM reads per launch -> {by/(sum(d)/len(d))/1e3:.0f} GB/s of traffic")
    facts["single_query"] = {"kernel": s1name, "kernel_mean_us": sum(d) / len(d), "FETCH_SIZE_KB_mean": f1, "hbm_read_bytes_per_launch": int(by)}
print()
# ---- flat ----
af = show_trace("cfg2 flat scan N=1M d=128 (scripts/bench_flat.py, 4 corpora in rotation)", "flat/trace", 4, top=5)
fname = "flat1h_kernel" if any("flat1h_kernel" in k for k in af) else "FlatSrc"   # (the corpus' fp16 shadow since round 6, else the f32 ordered chains)
ff, nf = mean_of(pmc("flat/pmc_fetch"), fname, "FETCH_SIZE", 4)
kf = [v for k, v in af.items() if (fname in k) and (fname == "flat1h_kernel" or "scan_kernel<1" in k)]
if kf and ff:
    d = kf[0][4:]
    alg = 1e6 * (128 * 2 + 4) if fname == "flat1h_kernel" else 512e6
    print(f"flat {fname}: mean {sum(d)/len(d):.1f} us (min {min(d):.1f}); algorithmic {alg/1e6:.1f} MB -> {alg/(sum(d)/len(d))/1e3:.0f} GB/s = "
          f"{alg/(sum(d)/len(d))/1e3/8000*100:.1f} % of 8 TB/s; FETCH_SIZE mean {ff:.0f} KB x 2 = {ff*1024*2/1e6:.0f} MB of HBM reads per launch")
    for nm in ("ivf_rescore_kernel<16>", "fallback_kernel", "flat_merge_kernel"):
        for k, v in af.items():
            if nm in k and len(v) > 4:
                print(f"  {short(k)[:70]:70s} mean {sum(v[4:])/len(v[4:]):6.1f} us over {len(v[4:])} warm dispatches")
    facts["flat_cfg2"] = {"kernel": fname, "kernel_mean_us": sum(d) / len(d), "FETCH_SIZE_KB_mean": ff, "algorithmic_bytes": alg}
print()
# ---- the headline configuration: three batches in flight (bench.py default) ----
a33 = trace("cfg3_s3/trace")
pk3 = [v for k, v in a33.items() if PK in k]
if pk3:
    d = pk3[0][5:25]   # the 20 timed launches (5 warm-up before them; the one-batch pass of the same run comes after)
    d1 = pk3[0][30:50]
    print(f"== cfg3, THREE batches in flight (bench.py default --streams 3): {PK}...> over the 20 timed launches: mean {sum(d)/len(d):.1f} us (min {min(d):.1f}) -- other batches' "
          f"coarse / planning / finish kernels run beside it; the same run's one-batch pass: mean {sum(d1)/max(1,len(d1)):.1f} us ==")
    facts["cfg3_s3"] = {"kernel_mean_us_timed_region": sum(d) / len(d), "kernel_mean_us_one_batch_pass": sum(d1) / max(1, len(d1))}
print()
# ---- d = 1536: the hi-only query block ----
ad = show_trace("d = 1536, N = 5M, nlist = 4096, batch 1024 (scripts/bench_d1536.py)", "d1536/trace", 5, top=6)
fd, nd = mean_of(pmc("d1536/pmc_fetch"), "prescan_kernel_g<true", "FETCH_SIZE", 5)
kd = [v for k, v in ad.items() if "prescan_kernel_g<true" in k]
if kd and fd:
    d = kd[0][5:]
    print(f"prescan_kernel_g<true, 32, IvfSrc<32>, false> (query block fp16 hi only): mean {sum(d)/len(d):.1f} us (min {min(d):.1f}) over {len(d)} warm dispatches; FETCH_SIZE mean {fd:.0f} KB x 2 = "
          f"{fd*1024*2/1e9:.2f} GB of HBM reads per launch -> {fd*1024*2/(sum(d)/len(d))/1e6:.2f} TB/s of traffic")
    try:
        print("  " + [l for l in open(os.path.join(out, "d1536", "trace.log")) if l.startswith("d=1536")][-1].strip())
    except (OSError, IndexError):
        pass
    facts["d1536"] = {"kernel_mean_us": sum(d) / len(d), "FETCH_SIZE_KB_mean": fd, "hbm_read_bytes_per_launch": int(fd * 1024 * 2)}
print()
# ---- the edges of the fast domain (round 6: wide candidate lists) ----
try:
    ae = show_trace("edges of the fast domain (scripts/bench_edges.py ONLY=headline,nprobe_128,nprobe_256,top_k_48,top_k_64,top_k_100,top_k_128; batch 256 but the headline; 3 warm-up + 8 timed calls per shape)", "edges/trace", 0, top=16)
    for l in open(os.path.join(out, "edges", "trace.log")):
        if l.startswith("{"):
            print("  " + l.strip()[:400])
except Exception as e:
    print("edges: not profiled", e)
print()
# ---- k-means assign contraction (scripts/bench_assign.py: N=4M k=4096 2 iterations, N=1M k=65536 1 iteration) ----
ak = show_trace("k-means builds (scripts/bench_assign.py)", "kmeans/trace", 0, top=12)
mk = pmc("kmeans/pmc_mfma")
kk = [(k, v) for k, v in ak.items() if "dist_gemm_x3w_kernel" in k or "dist_gemm_h_kernel" in k or "dist_gemm_x3_kernel<true" in k]
km_facts = {}
for name, v in kk:
    # launches of the first build (k = 4096: 2.5 ms each) and of the second (k = 65536: 40 ms each) are told apart by duration
    small = [x for x in v if x < 10000.0]; big = [x for x in v if x >= 10000.0]
    busy = mk.get(name, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", []); gui = mk.get(name, {}).get("GRBM_GUI_ACTIVE", [])
    for label, sel_, flops, key in (("k=4096 [131072 x 4096 x 768]", small, 2.0 * 131072 * 4096 * 768, "4096"), ("k=65536 [131072 x 65536 x 768]", big, 2.0 * 131072 * 65536 * 768, "65536")):
        if not sel_:
            continue
        med = sorted(sel_)[len(sel_) // 2]
        full = [x for x in sel_ if 0.8 * med < x < 1.25 * med]   # whole 131072-point batches (the probing batch of a pass and its last one are short; a cold first launch is long)
        mean = sum(full) / len(full)
        line = f"{short(name)[:60]} {label}: {len(full)} full launches, mean {mean:.1f} us = {flops / mean / 1e6:.1f} algorithmic TFLOP/s"
        if busy and gui and len(busy) == len(v):
            bsel = [b / 1024.0 / (g / 8.0) for b, g, x in zip(busy, gui, v) if (x < 10000.0) == (key == "4096")]
            if bsel:
                pct = 100.0 * sum(bsel) / len(bsel)
                line += f"; MFMA-busy {pct:.1f} % (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, PMC pass)"
                km_facts[key] = round(pct, 1)
        print(line)
if km_facts:
    json.dump({"kernel": "dist_gemm_h_kernel (k-means assign contraction from 4096 centroids on, 256 x 256 block tiles, persistent)", "mfma_busy_pct": km_facts,
               "source": "profiles/r06_summary.txt (rocprofv3 --pmc pass of scripts/bench_assign.py)"}, open(os.path.join(out, "kmeans.json"), "w"), indent=1)
print()
# ---- sharded search: every rank of every world (no profiler), then rank 0's per-kernel us from the traces ----
try:
    sj = json.load(open(os.path.join(out, "shard_all_ranks.json")))
    print("== sharded search, EVERY rank of every world measured on one GPU (scripts/emulate_shard.py 1 2 4 8; the step is ONE vers_ivf_search_sharded_dev call: partial search -> exchange STAND-IN WITH RCCL's FOOTPRINT (" + json.dumps(sj["config"].get("standin")) + ") -> merge of W partials; the scan's CU reserve: " + str(sj["config"].get("scan_reserve_cus")) + " (-1 = the library's auto policy)) ==")
    print(f"{'W':>2} {'rank':>4} {'stored rows':>12} {'probed rows':>12} {'step S=1 ms':>12} {'scan us':>8} {'step S=3 ms':>12} {'merge us':>9}")
    for w, v in sj["worlds"].items():
        for r in v["ranks"]:
            print(f"{w:>2} {r['rank']:>4} {r['stored_rows']:>12} {r['probed_rows']:>12} {r.get('step_ms_s1', float('nan')):>12.4f} {r.get('scan_us_s1', float('nan')):>8.1f} {r.get('step_ms_s3', float('nan')):>12.4f} {r['merge_us']:>9.1f}")
        print(f"   W={w}: step max / mean S=1 {v['step_ms_s1']['max']:.4f} / {v['step_ms_s1']['mean']:.4f} ms, S=3 {v['step_ms_s3']['max']:.4f} / {v['step_ms_s3']['mean']:.4f} ms (max / mean {v['step_ms_s3']['max_over_mean']}); probed rows max / mean {v['probed_rows']['max_over_mean']}")
    print("predicted speed-up from the SLOWEST rank's step (before the bytes of the peers travel):", {k: sj[k] for k in sj if k.startswith("predicted")})
    facts["shard_all_ranks"] = {w: {"step_ms_s1_max": v["step_ms_s1"]["max"], "step_ms_s3_max": v["step_ms_s3"]["max"], "step_ms_s3_mean": v["step_ms_s3"]["mean"],
                                    "probed_rows_max_over_mean": v["probed_rows"]["max_over_mean"]} for w, v in sj["worlds"].items()}
    facts["predicted_speedup"] = {k: sj[k] for k in sj if k.startswith("predicted")}
except (OSError, KeyError) as e:
    print("shard_all_ranks.json missing:", e)
print()
for W in (2, 4, 8):
    for tag, label, key in ((f"shard{W}/trace", "one batch in flight", "s1"), (f"shard{W}_s3/trace", "THREE batches in flight", "s3")):
        rows = []
        for f in glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        idx = [i for i, r in enumerate(rows) if "prescan_kernel" in r["Kernel_Name"]]
        if len(idx) < 30:
            continue
        # the timed steps of the run are its launches 7 .. 26 (6 warm-up steps first; 8 probed-rows calls and 50 merges follow): scan launch to scan launch
        a, b = idx[6], idx[26]
        span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
        agg = {}
        for r in rows[a:b]:
            k = short(r["Kernel_Name"])[:80]
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            x = agg.setdefault(k, [0, 0.0]); x[0] += 1; x[1] += d
        print(f"== rank 0 of {W}, {label}: 20 steps, scan launch to scan launch ==")
        print(f"wall per step {span/20:.1f} us; kernel time per step{' (kernels of different batches overlap)' if key == 's3' else ''} {sum(t for _, t in agg.values())/20:.1f} us")
        for k, (c, t) in agg.items():
            print(f"{t/20:9.1f} us/step  x{c/20:.1f}  {k}")
        facts[f"shard{W}_{key}"] = {"wall_us_per_step": span / 20, "kernel_us_per_step": {k: round(t / 20, 1) for k, (c, t) in agg.items()}}
        print()
for f in ("cfg4_rank_nlist16384.json", "cfg4_rank_nlist4096.json", "cfg5_rank.json"):
    try:
        r = json.load(open(os.path.join(out, f)))
    except OSError:
        continue
    print(f"== {f}: {r['workload']} ==")
    if "search" in r:
        s_ = r["search"]; u = r["upload"]
        print(f"   streamed upload kept {u['stored_rows']} rows of {r['rows_total']} in {u['seconds']} s; library {u['library_bytes_now']/1e9:.1f} GB (peak {u['library_bytes_peak']/1e9:.1f}), row-major copy kept {u['rowmajor_kept']}, shadow kept {u['shadow_kept']}")
        print(f"   step {s_['step_ms']} ms ({s_['step_includes']}); list scan alone {s_['list_scan_alone_ms']} ms = {s_['list_scan_frac_of_8TBs']} of 8 TB/s on {s_['algorithmic_bytes']/1e9:.2f} GB; re-scanned queries {s_['rescanned_queries']}")
        print(f"   GPU == CPU restatement bitwise over the rank's sub-index: {r.get('check', {}).get('gpu_matches_cpu_bitwise')} on {r.get('check', {}).get('queries')} queries; assign pass of the whole corpus {r['assign_pass']}")
    else:
        print(f"   build {r['build_s']} s; {r['assign_passes']} assign passes at {r['seconds_per_assign_pass']} s = {r['assign_pass_algorithmic_tflops']} algorithmic TFLOP/s (contraction alone {r['contraction_algorithmic_tflops']}); "
              f"library peak {r['library_bytes_peak']/1e9:.1f} GB = {r['peak_over_rows_bytes']} x the rows; properties {r['properties']}")
    facts[f[:-5]] = r
    print()
json.dump(facts, open(os.path.join(out, "facts.json"), "w"), indent=1)
