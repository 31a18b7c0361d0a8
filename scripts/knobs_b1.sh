#!/bin/bash
# Single-query list scan (cfg3, B=1, nprobe=32) against the segment size of its work items.  (on the GPU box)
for S in default 128 192 256 default; do
  if [ "$S" = default ]; then unset VERS_SEG_ROWS; else export VERS_SEG_ROWS=$S; fi
  echo -n "seg_rows=$S  "; python bench.py --batch 1 --steps 300 --warmup 20 --no-cpu --no-recall --kmeans-iters 2 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], 'ms/query  scan', r['launch_ms'], 'ms  frac', r['frac'], 'items', r['work_items'])"
done
