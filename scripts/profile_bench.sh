#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3: kernel trace + stats, then PMC passes (counters in
# their own runs, never combined with other trace domains).  Outputs land in gpurun_out/prof_<tag>/.
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu --no-recall $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_SMEM
pass fetch FETCH_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -60 "$OUT/summary.txt"
