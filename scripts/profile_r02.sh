#!/bin/bash
# Round-2 evidence for profiles/: kernel traces (warm: means reproduce the bench line) and PMC passes (own runs, never
# combined with other trace domains) of
#   cfg3 batch 1024   bench.py default            -> prescan_kernel_g, dist_gemm_kernel<false>, plan_fused, ivf_rescore
#   cfg3 single query bench.py --batch 1          -> scan_kernel<1,0,IvfSrc<1>>
#   cfg2 flat scan    scripts/bench_flat.py       -> scan_kernel<1,0,FlatSrc>
#   8-way shard       scripts/emulate_shard.py 8  -> per-kernel us of one rank's step
# usage (GPU box): bash scripts/profile_r02.sh ; outputs under gpurun_out/prof_r02/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r02
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B3="--steps 20 --warmup 5 --no-cpu --no-recall"   # (with the extra block: it also times the f32 MFMA kernel of the coarse contraction and the list scan on f32 rows)
B1="--batch 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2"
run() { # tag, counters ("" = kernel trace + stats), program args...
  local tag=$1 ctr=$2; shift 2
  mkdir -p "$OUT/$(dirname "$tag")"
  if [ -z "$ctr" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python3 "$@" > "$OUT/$tag.log" 2>&1
  else rocprofv3 --pmc $ctr --output-format csv -d "$OUT/$tag" -- python3 "$@" > "$OUT/$tag.log" 2>&1; fi
}
run cfg3/trace "" "$ROOT/bench.py" $B3
run cfg3/pmc_fetch "FETCH_SIZE" "$ROOT/bench.py" $B3
run cfg3/pmc_write "WRITE_SIZE" "$ROOT/bench.py" $B3
run cfg3/pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "$ROOT/bench.py" $B3
run cfg3/pmc_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "$ROOT/bench.py" $B3
run b1/trace "" "$ROOT/bench.py" $B1
run b1/pmc_fetch "FETCH_SIZE" "$ROOT/bench.py" $B1
run flat/trace "" "$ROOT/scripts/bench_flat.py"
run flat/pmc_fetch "FETCH_SIZE" "$ROOT/scripts/bench_flat.py"
run shard8/trace "" "$ROOT/scripts/emulate_shard.py" 8 0
python3 "$ROOT/scripts/summarize_r02.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
