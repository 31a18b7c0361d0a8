"""bench.py end to end on small shapes: the default single-GPU line carries the contract's keys, and `--gpus 2`
starts its own two ranks (sharing the one GPU of the test box through gloo: VERS_BENCH_BACKEND=gloo), builds the
index ROW-SHARDED and reports n_gpus = 2 with the same results (recall against the exact scan)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "20000", "--d", "64", "--nlist", "16", "--nprobe", "4", "--batch", "64", "--steps", "2", "--warmup", "1"]


def run_bench(*extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + list(extra), capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_bench_single_gpu_line():
    out, _ = run_bench("--cpu-seconds", "2")
    assert out["n_gpus"] == 1 and out["unit"] == "queries/sec" and out["value"] > 0
    for key in ("roofline", "cpu_baseline", "recall_at_10", "config", "ms_per_step", "scaling", "dtype"):
        assert key in out
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1.5
    assert out["cpu_baseline"]["gpu_matches_cpu_bitwise"] is True
    assert out["self_retrieval_ok"] is True
    assert out["cpu_baseline_all_cores"]["cores"] >= 1 and out["cpu_baseline_all_cores"]["gpu_matches_cpu_bitwise"] is True
    ex = out["extra"]
    assert ex["single_query"]["list_scan_us"] > 0 and ex["flat_cfg2"]["l2sq"]["frac"] > 0 and 0 <= ex["recall_at_10_dist_u"]["value"] <= 1
    # every leg the line promises is there (a leg that raises is logged, not dropped silently: the coarse contraction's entries
    # went missing from round 3's line for a while because the legs before it left a single query as the handle's last call)
    for key in ("coarse_gemm", "coarse_gemm_f32", "reference_mode", "list_scan_f32_rows", "memory", "batch_sweep", "domain_edges", "d1536", "build_index_phases_ms", "add"):  # (kmeans_assign: matrix-core builds only)
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
    assert "vers_ivf_search_sharded_dev" in two["config"]["exchange"]   # partial -> exchange -> merge inside ONE library call per batch


def test_bench_multi_rank_path_through_rccl_with_one_rank():
    """VERS_BENCH_FORCE_SHARDED=1: everything `bench.py --gpus N` runs on the GPUs -- the nccl process group, the row-sharded
    build entry, libvers_rccl.so's communicator made from an id broadcast through torch.distributed, vers_ivf_search_sharded_dev
    with ncclAllGather queued on the batch's stream, three batches in flight -- with the ONE rank a test box has (RCCL refuses
    two ranks on one GPU).  Same recall as the plain single-GPU run."""
    one, _ = run_bench("--no-cpu", "--no-extra")
    rc, err = run_bench("--no-cpu", "--no-extra", env={"VERS_BENCH_FORCE_SHARDED": "1"})
    assert rc["n_gpus"] == 1 and rc["recall_at_10"] == one["recall_at_10"] and rc["value"] > 0
    assert "libvers_rccl.so" in rc["config"]["exchange"] and "ncclAllGather" in rc["config"]["exchange"], rc["config"]["exchange"]
    assert "row-sharded build over 1 ranks" in err


def test_bench_gpus_8_over_gloo_on_one_gpu():
    """world 8 -- the size the driver's scaling run ends at -- through every rank-count-dependent piece of the run (LPT over 8
    owners, the row-sharded build's chain of 7 hops, partials of 8 ranks merged per batch) with the eight ranks sharing the test
    box's GPU over gloo; same recall as one rank."""
    one, _ = run_bench("--no-cpu", "--no-extra")
    eight, err = run_bench("--gpus", "8", "--no-cpu", "--no-extra", env={"VERS_BENCH_BACKEND": "gloo"})
    assert eight["n_gpus"] == 8 and eight["recall_at_10"] == one["recall_at_10"]
    assert "lists sharded over 8 ranks (LPT)" in err and "vers_ivf_search_sharded_dev" in eight["config"]["exchange"]
