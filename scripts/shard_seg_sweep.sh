#!/bin/bash
# Per-rank step of an N-way sharded search (scripts/emulate_shard.py) against the segment-size target of the
# matrix-core list scan.  usage: scripts/shard_seg_sweep.sh  (on the GPU box)
for W in 8 4; do
  for S in default 128 192 384 640; do
    if [ "$S" = default ]; then unset VERS_SEG_ROWS; else export VERS_SEG_ROWS=$S; fi
    echo -n "seg_rows=$S  "; timeout 300 python scripts/emulate_shard.py $W 0 2>&1 | tail -1
  done
done
