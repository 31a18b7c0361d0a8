#!/bin/bash
# in-process-ish A/B of the batched list scan (same box, alternating): VERS_SCAN_DEBUG bit 3 (8) = no shared
# pruning bounds, bit 5 (32) = static quad stride, bit 6 (64) = single-pair dead-slot granularity
for rep in 1 2; do for f in ${AB_FLAGS:-0 64}; do
  VERS_SCAN_DEBUG=$f python bench.py --steps 8 --warmup 2 --no-cpu --no-recall --kmeans-iters 2 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$f scan ms', d['roofline']['launch_ms'], 'qps', d['value'])"
done; done
