#!/bin/bash
# Round-6 evidence for profiles/: kernel traces (warm: means reproduce the bench line) and PMC passes (own runs, never combined
# with other trace domains) of
#   cfg3 batch 1024, one batch in flight   bench.py --streams 1        -> prescan_kernel_g, coarse kernels, group_scatter, ivf_rescore
#   cfg3 batch 1024, bench default         bench.py (3 in flight)      -> the same launches with other batches beside them
#   cfg3 single query                      bench.py --batch 1          -> coarse1_kernel, scan1h_kernel, ivf_rescore_kernel<16>, fallback_kernel
#   cfg2 flat scan                         scripts/bench_flat.py       -> scan_kernel<1,0,FlatSrc>
#   k-means assign                         scripts/bench_assign.py     -> dist_gemm_x3w_kernel<2, 1> (the cascade's fp16 single product) (+ MFMA-busy by PMC)
#   the edges of the fast domain           scripts/bench_edges.py      -> prescan_kernel_g<.., WIDE>, ivf_rescore_wide_kernel, coarse_select_wide_kernel
#   d = 1536                               scripts/bench_d1536.py      -> prescan_kernel_g<true, 32, .., LO = false> (query block as fp16 hi only)
#   sharded search, EVERY rank of W = 1, 2, 4, 8 (no profiler)         -> shard_all_ranks.json: per-rank step, max / mean, probed rows;
#                                                                         the exchange is the stand-in with RCCL's footprint, the scan's CU reserve AUTO
#   cfg4 / cfg5 at one rank's nominal size (no profiler)               -> cfg4_rank_nlist16384.json, cfg4_rank_nlist4096.json, cfg5_rank.json
#   rank 0 of W = 2, 4, 8: kernel traces, one and three batches in flight -> per-kernel us of one rank's step
# usage (GPU box): bash scripts/profile_r06.sh <commit>   ; outputs under gpurun_out/prof_r06/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r06
rm -rf "$OUT"; mkdir -p "$OUT"
echo "${1:-unknown}" > "$OUT/commit.txt"
sha256sum "$ROOT/vers_amd/lib/libvers_hip.so" | cut -c1-16 > "$OUT/lib_sha16.txt"
cd /tmp && export TMPDIR=/tmp
B3="--streams 1 --steps 20 --warmup 5 --no-cpu --no-recall --no-extra"
B1="--batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2"
run() { # tag, counters ("" = kernel trace + stats), program args...
  local tag=$1 ctr=$2; shift 2
  mkdir -p "$OUT/$(dirname "$tag")"
  if [ -z "$ctr" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python3 "$@" > "$OUT/$tag.log" 2>&1
  else rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$tag" -- python3 "$@" > "$OUT/$tag.log" 2>&1; fi
}
# (the all-ranks table first, without a profiler)
OUT_JSON=$OUT/shard_all_ranks.json
( cd "$ROOT" && EMU_OUT=$OUT_JSON python3 scripts/emulate_shard.py 1 2 4 8 > "$OUT/shard_all_ranks.log" 2>&1 )
( cd "$ROOT" && python3 scripts/rank_nominal.py cfg4 NLIST=16384 OUT=$OUT/cfg4_rank_nlist16384.json > "$OUT/cfg4_rank_nlist16384.log" 2>&1 )
( cd "$ROOT" && python3 scripts/rank_nominal.py cfg4 NLIST=4096 OUT=$OUT/cfg4_rank_nlist4096.json > "$OUT/cfg4_rank_nlist4096.log" 2>&1 )
( cd "$ROOT" && python3 scripts/rank_nominal.py cfg5 OUT=$OUT/cfg5_rank.json > "$OUT/cfg5_rank.log" 2>&1 )
run cfg3/trace "" "$ROOT/bench.py" $B3                                  # (clean: ONLY the timed configuration's launches -- ADVICE r04)
run cfg3x/trace "" "$ROOT/bench.py" --streams 1 --steps 20 --warmup 5 --no-cpu --no-recall --no-nominal   # (the extras' kernels: f32 coarse contraction, f32-row scan, ... in their own pass)
run cfg3/pmc_fetch "FETCH_SIZE" "$ROOT/bench.py" $B3
run cfg3/pmc_write "WRITE_SIZE" "$ROOT/bench.py" $B3
run cfg3/pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "$ROOT/bench.py" --streams 1 --steps 20 --warmup 5 --no-cpu --no-recall --no-nominal
run cfg3/pmc_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "$ROOT/bench.py" $B3
run cfg3_s3/trace "" "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu --no-recall --no-extra
run b1/trace "" "$ROOT/bench.py" $B1
run b1/pmc_fetch "FETCH_SIZE" "$ROOT/bench.py" $B1
run flat/trace "" "$ROOT/scripts/bench_flat.py"
run flat/pmc_fetch "FETCH_SIZE" "$ROOT/scripts/bench_flat.py"
run d1536/trace "" "$ROOT/scripts/bench_d1536.py"
run d1536/pmc_fetch "FETCH_SIZE" "$ROOT/scripts/bench_d1536.py"
run edges/trace "" "$ROOT/scripts/bench_edges.py" ONLY=headline,nprobe_128,nprobe_256,top_k_48,top_k_64,top_k_100,top_k_128   # (round 6: wide lists -- prescan WIDE, ivf_rescore_wide_kernel, coarse_select_wide_kernel)
run kmeans/trace "" "$ROOT/scripts/bench_assign.py"
run kmeans/pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "$ROOT/scripts/bench_assign.py"
for W in 2 4 8; do
  RANKS=0 STREAMS=1 EMU_OUT=$OUT/shard${W}_s1.json run shard${W}/trace "" "$ROOT/scripts/emulate_shard.py" $W
  RANKS=0 STREAMS=3 EMU_OUT=$OUT/shard${W}_s3.json run shard${W}_s3/trace "" "$ROOT/scripts/emulate_shard.py" $W
done
python3 "$ROOT/scripts/summarize_r06.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
