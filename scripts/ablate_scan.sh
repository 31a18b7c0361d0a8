#!/bin/bash
# Ablation of the batched list scan on hardware (diagnosis only): VERS_SCAN_DEBUG bit 0 skips the top-k
# fold, bit 1 the distance math, bit 2 reads one query column only.
for f in 0 1 2 3 4; do
  VERS_SCAN_DEBUG=$f python bench.py --steps 4 --warmup 1 --no-cpu --no-recall --kmeans-iters 2 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$f scan ms', d['roofline']['launch_ms'], 'streamed GB', round(d['roofline']['streamed_bytes_per_launch']/1e9,1), 'qps', d['value'], 'items', d['roofline']['work_items'])"
done
