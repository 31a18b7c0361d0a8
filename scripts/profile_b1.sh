#!/bin/bash
# the single-query section only: kernel trace + FETCH_SIZE pass of bench.py --batch 1 -> gpurun_out/prof_r03/b1 (re-taken on the
# final binary; delete the local gpurun_out/prof_r03/b1 first: the merge-back adds files, it does not remove old ones)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p "$OUT/b1"; rm -rf "$OUT/b1/trace" "$OUT/b1/pmc_fetch"
cd /tmp && export TMPDIR=/tmp
B1="--batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b1/trace" -- python3 "$ROOT/bench.py" $B1 > "$OUT/b1/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/b1/pmc_fetch" -- python3 "$ROOT/bench.py" $B1 > "$OUT/b1/pmc_fetch.log" 2>&1
python3 "$ROOT/scripts/timeline_b1.py" "$OUT/b1/trace"
