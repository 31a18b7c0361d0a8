#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_b1
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --batch 1 --steps 200 --warmup 10 --no-cpu --no-recall --kmeans-iters 2 > "$OUT/trace.log" 2>&1
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" | head -30
