#!/bin/bash
# k-means assign contraction only: kernel trace + MFMA-busy PMC pass of scripts/bench_assign.py -> gpurun_out/prof_r03/kmeans (re-taken on the final binary)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p "$OUT/kmeans"; rm -rf "$OUT/kmeans/trace" "$OUT/kmeans/pmc_mfma"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kmeans/trace" -- python3 "$ROOT/scripts/bench_assign.py" > "$OUT/kmeans/trace.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/kmeans/pmc_mfma" -- python3 "$ROOT/scripts/bench_assign.py" > "$OUT/kmeans/pmc_mfma.log" 2>&1
python3 "$ROOT/scripts/summarize_r03.py" "$OUT" 2>/dev/null | grep "dist_gemm_x3w"
