#!/bin/bash
# Same-box A/B of the batched list scan at cfg3: ordered-chain scan (VERS_PRESCAN=0) vs matrix cores (default),
# optional seg-row sweep for the matrix-core kernel: AB_SEGS="0 256 512 1024" (0 = default geometry)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
run() { # $1 = VERS_PRESCAN, $2 = VERS_SEG_ROWS or 0
  local extra=""
  [ "$2" != "0" ] && extra="VERS_SEG_ROWS=$2"
  env VERS_PRESCAN=$1 $extra python bench.py --steps 8 --warmup 2 --no-cpu --no-recall --kmeans-iters 2 2>gpurun_out/ab_err.log | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('prescan=$1 seg=$2 scan ms', r['launch_ms'], 'frac', r['frac'], 'qps', d['value'], 'self_ok', d['self_retrieval_ok'], 'items', r['work_items'])"
  grep -E "matrix cores:|GPU == CPU|mismatch" gpurun_out/ab_err.log | tail -2
}
for rep in 1 2; do
  run 0 0
  for s in ${AB_SEGS:-0}; do run 1 $s; done
done
