"""The ONE stdout line of bench.py: the contract's keys + numbers of the legs, at most LINE_LIMIT characters (the driver keeps the last
8,000 characters of stdout: round 5's 20.7 KB line was lost).  The full result -- every leg with its prose -- goes to a side file and,
leg by leg, to stderr."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


LINE_LIMIT = 4000   # characters of the ONE stdout line (the driver keeps the last 8,000 characters of stdout: round 5's 20.7 KB line was lost)


def _g(dct, *path, default=None):
    """dct[path[0]][path[1]]... or `default` when any level is missing (a leg that was skipped or failed)"""
    for k in path:
        if not isinstance(dct, dict) or k not in dct:
            return default
        dct = dct[k]
    return dct


def compact_line(out: dict) -> dict:
    """The stdout line: the contract's keys + NUMBERS of the legs.  Every string of prose (kernel names, notes, samples) and the
    full leg dictionaries stay in the side file / on stderr (emit)."""
    rf, ex = out["roofline"], out.get("extra") or {}
    shadow = str(rf.get("row_operand", "")).startswith("fp16")
    f32k = ex.get("list_scan_f32_rows") or {}
    traffic = rf.get("traffic")
    roofline = {"bound": rf["bound"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"], "traffic": traffic,
                "traffic_over_algorithmic": None if not traffic else round(traffic / max(1, rf["algorithmic_bytes_per_launch"]), 4),
                "kernel": str(rf["kernel"]).split(" (")[0], "row_operand": "fp16" if shadow else "f32", "bytes_per_element": 2 if shadow else 4,
                "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "launch_ms": rf["launch_ms"], "launches_timed": rf["launches_timed"],
                # SURVEY 8d's own denominator (4 B per element of the union of probed lists): the kernel that streams the f32 rows, same steps
                "frac_f32_rows_kernel": f32k.get("frac"), "launch_ms_f32_rows_kernel": f32k.get("launch_ms"),
                "f32_rows_bytes_per_launch": _g(rf, "f32_rows_equivalent", "bytes_per_launch", default=rf["algorithmic_bytes_per_launch"] if not shadow else None),
                "timed_region_launch_ms": _g(rf, "timed_region", "launch_ms"), "one_in_flight_step_ms": _g(rf, "one_batch_in_flight", "whole_step_ms")}
    cpu = out.get("cpu_baseline")
    if cpu:
        cpu = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
               "sample": str(cpu["sample"]).split(":")[0] + ", one oracle call each", "gpu_matches_cpu_bitwise": cpu["gpu_matches_cpu_bitwise"]}
    cfg = out["config"]
    config = {k: cfg[k] for k in ("n", "d", "nlist", "nprobe", "batch", "top_k", "kmeans_iters", "batches_in_flight") if k in cfg}
    config["workload"] = f"IVFFlat N={cfg['n']} d={cfg['d']} nlist={cfg['nlist']} nprobe={cfg['nprobe']} batch={cfg['batch']} top_k={cfg['top_k']} f32 Dist-C"
    config["parallelism"] = f"lists sharded over {out['n_gpus']} GPU(s)"
    config["exchange"] = cfg.get("exchange_tag")
    sq, fl, sw, ed = ex.get("single_query") or {}, ex.get("flat_cfg2") or {}, _g(ex, "batch_sweep", "by_batch", default={}), _g(ex, "domain_edges", "by_shape", default={})
    legs = {
        "single_query": {k: sq.get(k) for k in ("list_scan_us", "frac", "end_to_end_us", "host_call_us")} if sq else None,
        "reference_mode": {k: _g(ex, "reference_mode", k) for k in ("batch_queries_per_sec", "single_query_end_to_end_us", "single_query_host_call_us", "gpu_matches_cpu_bitwise")} if "reference_mode" in ex else None,
        "flat_cfg2": {"scan_us": _g(fl, "l2sq", "scan_us"), "frac": _g(fl, "l2sq", "frac"), "queries_per_sec": fl.get("queries_per_sec"),
                      "shadow_scan_us": _g(fl, "on_the_shadow", "l2sq", "scan_us"), "shadow_frac": _g(fl, "on_the_shadow", "l2sq", "frac"),
                      "shadow_queries_per_sec": _g(fl, "on_the_shadow", "queries_per_sec"), "cpu_queries_per_sec": _g(fl, "cpu_baseline", "value"),
                      "gpu_matches_cpu_bitwise": _g(fl, "cpu_baseline", "gpu_matches_cpu_bitwise")} if fl else None,
        "coarse_gemm": {"us": _g(ex, "coarse_gemm", "us"), "tflops": _g(ex, "coarse_gemm", "algorithmic_tflops"), "f32_mfma_us": _g(ex, "coarse_gemm_f32", "us"),
                        "f32_mfma_frac": _g(ex, "coarse_gemm_f32", "frac")} if "coarse_gemm" in ex else None,
        "kmeans_assign": {"tflops": _g(ex, "kmeans_assign", "algorithmic_tflops"), "frac_f16_dense": _g(ex, "kmeans_assign", "frac_of_f16_dense"), "assign_pass_ms": _g(ex, "kmeans_assign", "assign_pass_ms"),
                          "mfma_busy_pct": _g(ex, "kmeans_assign", "mfma_busy_pct"), "k65536_tflops": _g(ex, "kmeans_assign_k65536", "algorithmic_tflops")} if "kmeans_assign" in ex else None,
        "batch_sweep_qps": {k: v.get("queries_per_sec") for k, v in sw.items()} or None,
        "batch_sweep_bitwise": all(v.get("gpu_matches_cpu_bitwise", True) for v in sw.values()) if sw else None,
        "domain_edges_qps": {k: v.get("queries_per_sec") for k, v in ed.items()} or None,
        "d1536": {k: _g(ex, "d1536", k) for k in ("queries_per_sec", "frac", "gpu_matches_cpu_bitwise")} if "d1536" in ex else None,
        "memory": {"bytes_per_row": _g(ex, "memory", "bytes_per_stored_row"), "over_f32_rows": None if not _g(ex, "memory", "bytes_per_stored_row") else
                   round(_g(ex, "memory", "bytes_per_stored_row") / (4.0 * cfg["d"]), 3), "max_N_per_gpu": _g(ex, "memory", "max_N_per_gpu_as_configured"),
                   "compact_over_f32_rows": _g(ex, "memory_compact", "over_f32_rows"), "compact_step_ms": _g(ex, "memory_compact", "whole_step_ms"),
                   "compact_step_ms_one_in_flight": _g(ex, "memory_compact", "one_batch_in_flight_whole_step_ms"), "compact_same_results": _g(ex, "memory_compact", "same_results_as_the_headline_bitwise")} if "memory" in ex else None,
        "build_index_s": ex.get("build_index_s"), "add_us": _g(ex, "add", "us_per_vector"),
        "recall_dist_u": _g(ex, "recall_at_10_dist_u", "value"), "recall_vs_nprobe": _g(ex, "recall_vs_nprobe_noisy_dist_c", "recall_at_10"),
    }
    for name in ("cfg4_rank", "cfg4_rank_nlist4096"):
        r = ex.get(name)
        if r:
            legs[name] = {"failed": True} if "failed" in r else {
                "step_ms_s1": _g(r, "search", "step_ms", "s1"), "step_ms_s3": _g(r, "search", "step_ms", "s3"), "scan_frac": _g(r, "search", "list_scan_frac_of_8TBs"),
                "gpu_matches_cpu_bitwise": _g(r, "check", "gpu_matches_cpu_bitwise"), "library_gb": None if not _g(r, "upload", "library_bytes_now") else round(_g(r, "upload", "library_bytes_now") / 1e9, 1)}
    r = ex.get("cfg5_rank")
    if r:
        legs["cfg5_rank"] = {"failed": True} if "failed" in r else {"s_per_assign_pass": r.get("seconds_per_assign_pass"), "tflops": r.get("contraction_algorithmic_tflops"),
                                                                     "peak_over_rows": r.get("peak_over_rows_bytes"), "properties_ok": all(v for v in (r.get("properties") or {}).values())}
    km = out.get("cpu_baseline_kmeans") or {}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                                "recall_at_10", "self_retrieval_ok")}
    line["config"] = config
    line["roofline"] = roofline
    line["cpu_baseline"] = cpu
    line["value_f32_rows"] = out.get("value_f32_rows")
    line["row_operand"] = roofline["row_operand"] + (" pre-filter, exact f32 finish in the step" if shadow else "")
    ca = out.get("cpu_baseline_all_cores")
    line["cpu_baseline_all_cores"] = None if not ca else {"value": ca["value"], "cores": ca["cores"], "gpu_matches_cpu_bitwise": ca["gpu_matches_cpu_bitwise"]}
    line["cpu_kmeans"] = None if not km else {"assign_gflops": _g(km, "assign_to_clusters", "gflops_2nkd"), "cores": _g(km, "assign_to_clusters", "cores"),
                                              "assign_matches": _g(km, "assign_to_clusters", "gpu_matches_cpu"), "update_bitwise": _g(km, "update_centroids", "gpu_matches_cpu_bitwise"),
                                              "cost_bitwise": _g(km, "calculate_kmeans_cost", "gpu_matches_cpu_bitwise")}
    line["extra"] = {k: v for k, v in legs.items() if v is not None}
    line["extra_file"] = out.get("extra_file")
    return line


def emit(out: dict, extra_file: str | None):
    """Full result -> side file + one `[bench-extra] <leg> {json}` line per leg on stderr; the compact line -> stdout (the ONLY thing there)."""
    path = extra_file or os.path.join(ROOT, "gpurun_out", f"bench_extra_n{out['n_gpus']}.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        out["extra_file"] = os.path.relpath(path, ROOT)
    except OSError as e:
        log(f"[bench] could not write {path}: {e}")
        out["extra_file"] = None
    for k in ("roofline", "cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_kmeans", "config"):
        log(f"[bench-extra] {k} {json.dumps(out.get(k))}")
    for k, v in (out.get("extra") or {}).items():
        log(f"[bench-extra] extra.{k} {json.dumps(v)}")
    line = compact_line(out)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:   # (never reached with today's legs: ~2.5 KB; a future leg must not push the contract's keys out of the driver's buffer)
        log(f"[bench] compact line is {len(text)} characters > {LINE_LIMIT}: dropping extra")
        line["extra"] = {"dropped": True}
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    print(text, flush=True)
