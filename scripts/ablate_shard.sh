#!/bin/bash
# same-box ablation of the list scan at 8 ranks (rank 0) and on one GPU: what the launch costs without list inserts / math / both,
# with the hot-lists-first order off, and with build variants (VERS_LIB_PATH).  Results are WRONG with VERS_SCAN_DEBUG set: timing only.
# usage (GPU box): scripts/ablate_shard.sh [variant tags ...]
cd "$(dirname "$0")/.."
export RANKS=0 EXCHANGE=0 STREAMS=${STREAMS:-1,3}
run() { echo "== $1"; shift; env "$@" python scripts/emulate_shard.py 1 8 2>&1 | grep '"world"' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   W=%d' % j['world'], {k: j[k] for k in j if k.startswith('step') or k.startswith('scan')})"; }
for rep in 1 2; do
run "default (rep $rep)" A=1
for tag in "$@"; do run "variant $tag (rep $rep)" VERS_LIB_PATH=$PWD/vers_amd/lib/variants/libvers_hip_$tag.so; done
done
run "no list inserts (debug 1)" VERS_SCAN_DEBUG=1
run "no math (debug 2)" VERS_SCAN_DEBUG=2
run "no inserts, no math (debug 3)" VERS_SCAN_DEBUG=3
