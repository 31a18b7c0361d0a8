#!/bin/bash
# A/B of the batched list scan: VERS_SCAN_DEBUG bit 3 (8) = no shared pruning bounds, bit 5 (32) = static quad stride
for f in 0 8 32 40; do
  VERS_SCAN_DEBUG=$f python bench.py --steps 6 --warmup 2 --no-cpu --no-recall --kmeans-iters 2 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$f scan ms', d['roofline']['launch_ms'], 'qps', d['value'])"
done
