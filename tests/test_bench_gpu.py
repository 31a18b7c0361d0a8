"""bench.py end to end on small shapes: the default single-GPU line carries the contract's keys, and `--gpus 2`
starts its own two ranks (sharing the one GPU of the test box through gloo: VERS_BENCH_BACKEND=gloo), builds the
index ROW-SHARDED and reports n_gpus = 2 with the same results (recall against the exact scan)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "20000", "--d", "64", "--nlist", "16", "--nprobe", "4", "--batch", "64", "--steps", "2", "--warmup", "1"]


def run_bench(*extra, env=None, full=False):
    """-> (the stdout line, stderr) or, with full=True, (the line, the side file's full result, stderr).  The line is what the
    driver parses: ONE line, far below the 8,000 characters of stdout the driver keeps (round 5's 20.7 KB line was lost)."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    with tempfile.TemporaryDirectory() as td:
        side = os.path.join(td, "extra.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--extra-file", side] + list(extra), capture_output=True, text=True, timeout=900, env=e)
        assert r.returncode == 0, r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout     # nothing but the line on stdout
        assert len(lines[0]) < 4000, len(lines[0])
        line = json.loads(r.stdout[-8000:])                                 # what survives the driver's tail buffer parses on its own
        assert line == json.loads(lines[0])
        whole = json.load(open(side))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "workload" in line["config"] and "model" not in line["config"]
    return (line, whole, r.stderr) if full else (line, r.stderr)


def test_bench_single_gpu_line():
    line, out, _ = run_bench("--cpu-seconds", "2", full=True)
    assert line["n_gpus"] == 1 and line["unit"] == "queries/sec" and line["value"] > 0 and line["value"] == out["value"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1.5 and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["achieved"] > 0
    # both fractions side by side: the bytes the dominant kernel streams (2 B / element of the fp16 shadow) and SURVEY 8d's f32 rows
    assert rf["row_operand"] == "fp16" and rf["bytes_per_element"] == 2 and 0 < rf["frac_f32_rows_kernel"] < 1.5 and line["value_f32_rows"] > 0
    assert line["cpu_baseline"]["gpu_matches_cpu_bitwise"] is True and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert line["cpu_baseline"]["value"] > 0 and isinstance(line["cpu_baseline"]["sample"], str)
    assert line["self_retrieval_ok"] is True
    assert line["cpu_baseline_all_cores"]["cores"] >= 1 and line["cpu_baseline_all_cores"]["gpu_matches_cpu_bitwise"] is True
    lx = line["extra"]
    assert lx["single_query"]["end_to_end_us"] > 0 and lx["flat_cfg2"]["frac"] > 0 and lx["batch_sweep_bitwise"] is True and lx["d1536"]["gpu_matches_cpu_bitwise"] is True
    # the line carries numbers, not prose: no string longer than a short label anywhere in it
    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(s) for s in strings(line)) <= 100
    assert out["roofline"]["bound"] == "hbm" and out["cpu_baseline"]["gpu_matches_cpu_bitwise"] is True
    ex = out["extra"]
    assert ex["single_query"]["list_scan_us"] > 0 and ex["flat_cfg2"]["l2sq"]["frac"] > 0 and 0 <= ex["recall_at_10_dist_u"]["value"] <= 1
    # every leg the line promises is there (a leg that raises is logged, not dropped silently: the coarse contraction's entries
    # went missing from round 3's line for a while because the legs before it left a single query as the handle's last call)
    for key in ("coarse_gemm", "coarse_gemm_f32", "reference_mode", "list_scan_f32_rows", "memory", "memory_compact", "batch_sweep", "domain_edges", "d1536", "build_index_phases_ms", "add"):  # (kmeans_assign: matrix-core builds only)
        assert key in ex, (key, sorted(ex))
    assert all(v["gpu_matches_cpu_bitwise"] for v in ex["batch_sweep"]["by_batch"].values()) and ex["d1536"]["gpu_matches_cpu_bitwise"] is True
    assert "prescan_kernel_g<true, 32, IvfSrc<32>, LO = false>" in ex["d1536"]["list_scan"] and ex["d1536"]["queries_compared_bitwise"] >= 2 and out["row_operand"]
    assert ex["coarse_gemm"]["us"] > 0 and ex["coarse_gemm_f32"]["us"] > 0 and ex["single_query"]["end_to_end_us"] > 0


def test_bench_gpus_2_spawns_ranks_and_builds_row_sharded():
    one, _ = run_bench("--no-cpu", "--no-extra")
    two, err = run_bench("--gpus", "2", "--no-cpu", "--no-extra", env={"VERS_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2
    assert "row-sharded build over 2 ranks: 10000 rows generated per rank" in err
    assert two["recall_at_10"] == one["recall_at_10"]
    assert two["config"]["kmeans_iters"] == one["config"]["kmeans_iters"]
    assert two["config"]["exchange"] == "gloo_gather_inside_search_sharded_dev"   # partial -> exchange -> merge inside ONE library call per batch


def test_bench_multi_rank_path_through_rccl_with_one_rank():
    """VERS_BENCH_FORCE_SHARDED=1: everything `bench.py --gpus N` runs on the GPUs -- the nccl process group, the row-sharded
    build entry, libvers_rccl.so's communicator made from an id broadcast through torch.distributed, vers_ivf_search_sharded_dev
    with ncclAllGather queued on the batch's stream, three batches in flight -- with the ONE rank a test box has (RCCL refuses
    two ranks on one GPU).  Same recall as the plain single-GPU run."""
    one, _ = run_bench("--no-cpu", "--no-extra")
    rc, err = run_bench("--no-cpu", "--no-extra", env={"VERS_BENCH_FORCE_SHARDED": "1"})
    assert rc["n_gpus"] == 1 and rc["recall_at_10"] == one["recall_at_10"] and rc["value"] > 0
    assert rc["config"]["exchange"] == "rccl_allgather_inside_search_sharded_dev", rc["config"]["exchange"]
    assert "libvers_rccl.so" in err
    assert "row-sharded build over 1 ranks" in err


def test_bench_gpus_8_over_gloo_on_one_gpu():
    """world 8 -- the size the driver's scaling run ends at -- through every rank-count-dependent piece of the run (LPT over 8
    owners, the row-sharded build's chain of 7 hops, partials of 8 ranks merged per batch) with the eight ranks sharing the test
    box's GPU over gloo; same recall as one rank."""
    one, _ = run_bench("--no-cpu", "--no-extra")
    eight, err = run_bench("--gpus", "8", "--no-cpu", "--no-extra", env={"VERS_BENCH_BACKEND": "gloo"})
    assert eight["n_gpus"] == 8 and eight["recall_at_10"] == one["recall_at_10"]
    assert "lists sharded over 8 ranks (LPT)" in err and eight["config"]["exchange"] == "gloo_gather_inside_search_sharded_dev"
