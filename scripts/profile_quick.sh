#!/bin/bash
# quick PMC look at the list-scan kernel (k-means cut to 1 iteration; not the judged profile)
set -u
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu --no-recall --kmeans-iters 1 $*"
pass() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -A12 "IvfSrc" "$OUT/summary.txt"
