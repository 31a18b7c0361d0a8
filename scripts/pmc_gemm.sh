#!/bin/bash
# PMC look at the k-means assign contraction (dist_gemm_x3w_kernel): where its wave cycles go.  usage (GPU box): pmc_gemm.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp ONLY=0
pass() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/scripts/bench_assign.py" > "$OUT/$name.log" 2>&1; }
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + '/[abc]')):
    cnt = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'dist_gemm_x3w' in r['Kernel_Name']: cnt[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in sorted(cnt.items()): print('%-34s %14.5g  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
