#!/bin/bash
# kernel trace of the single-query chain at cfg3 + its per-kernel timeline (on the GPU box); extra env passes through
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_b1${TAG:-}
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2 > "$OUT/trace.log" 2>&1
python3 "$ROOT/scripts/timeline_b1.py" "$OUT/trace" "${1:-scan1_kernel}" | tee "$OUT/timeline.txt"
