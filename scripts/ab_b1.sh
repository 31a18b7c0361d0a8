#!/bin/bash
# single query at cfg3: coarse quantiser + plan in one launch (coarse1_kernel) against the ordered-chain scan + plan1_kernel, same box
B1="--batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2"
for v in 0 1 0 1; do
  echo -n "VERS_COARSE1=$v "; VERS_COARSE1=$v python bench.py $B1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], 'ms/query  scan', r['launch_ms'], 'frac', r['frac'])"
done
