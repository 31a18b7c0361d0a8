#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_b1ref
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --batch 1 --nprobe ${NPROBE:-0} --steps 200 --warmup 10 --no-cpu --no-recall --kmeans-iters 2 > "$OUT/trace.log" 2>&1
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" | grep -v "AssignSrc\|cost_fold\|row_norm\|gen_raw\|row_scale\|update_kernel\|rocprim\|count_kernel\|gather_rows\|dist_gemm\|assign_\|gather_points\|differs\|blocked_row" | head -16
tail -1 "$OUT/trace.log" | cut -c1-260
