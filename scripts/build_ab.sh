#!/bin/bash
# Same-box A/B of build_index at cfg3: exact-scan assign vs matrix-core assign (must give the same fingerprint).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for mode in 1 0; do
  echo "== VERS_ASSIGN=$mode" >> gpurun_out/build_ab.log
  VERS_ASSIGN=$mode timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-recall 2>&1 | grep -E "build_index|metric" >> gpurun_out/build_ab.log
done
cat gpurun_out/build_ab.log
