#!/bin/bash
# same-box A/B of library builds on one rank's step of a W-way sharded search: per-kernel us per step from a kernel trace
# usage (GPU box): ab_trace_shard.sh <W> <tag> [<tag> ...]    ("default" = the product build; others: vers_amd/lib/variants/libvers_hip_<tag>.so)
cd "$(dirname "$0")/.."
W=$1; shift
for tag in "$@"; do
  if [ "$tag" = default ]; then unset VERS_LIB_PATH; else export VERS_LIB_PATH=$PWD/vers_amd/lib/variants/libvers_hip_$tag.so; fi
  echo "== $tag (W=$W)"
  bash scripts/trace_shard.sh $W 2>&1 | tail -14
done
