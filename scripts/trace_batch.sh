#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_batch
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --no-recall --kmeans-iters 1 > "$OUT/trace.log" 2>&1
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" | grep -v "AssignSrc\|cost_fold\|row_norm_kernel\|gen_raw\|row_scale\|update_kernel\|rocprim\|count_kernel\|gather_rows" | head -24
tail -1 "$OUT/trace.log" | cut -c1-200
