#!/bin/bash
# Same-box knob sweep of the matrix-core list scan at cfg3: each argument is an "ENV=VAL[,ENV=VAL]" set ("-" = defaults)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for rep in ${REPS:-1 2}; do for kv in "$@"; do
  envs=""; [ "$kv" != "-" ] && envs=$(echo "$kv" | tr ',' ' ')
  env $envs python bench.py --steps 8 --warmup 2 --no-cpu --no-recall --kmeans-iters 2 2>gpurun_out/knob_err.log | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$kv: scan ms', r['launch_ms'], 'frac', r['frac'], 'qps', d['value'], 'step ms', d['ms_per_step'], 'streamed GB', round(r['streamed_bytes_per_launch']/1e9,2), 'items', r['work_items'])"
  grep -E "matrix cores:|vers stamps" gpurun_out/knob_err.log | tail -3
done; done
