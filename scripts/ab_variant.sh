#!/bin/bash
# same-box A/B of compile-time variants of the library (scripts/build_variant.sh) on the single-query path at cfg3
# usage (GPU box): ab_variant.sh <tag> [<tag> ...]   ("default" = the product build); extra env passes through
B1="--batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2"
for rep in 1 2; do
for tag in "$@"; do
  if [ "$tag" = default ]; then unset VERS_LIB_PATH; else export VERS_LIB_PATH=$PWD/vers_amd/lib/variants/libvers_hip_$tag.so; fi
  echo -n "$tag: "; python bench.py $B1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], 'ms/query  scan', r['launch_ms'], 'frac', r['frac'])"
done
done
