#!/bin/bash
# PMC look at the k-means assign contraction (k = 65536): usage scripts/pmc_assign.sh  -> gpurun_out/pmc_assign/*.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_assign; mkdir -p $OUT
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  ONLY=1 VERS_OPTIONS=assign_glds=1 rocprofv3 --pmc $ctr --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/bench_assign.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' > $OUT/p$i.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "gemm" in k:
        print(k, {c: (round(v / max(1, cnt[(k, c)]), 1)) for c, v in agg[k].items()}, "dispatches", max(cnt[(k, c)] for c in agg[k]))
PY
  cat $OUT/p$i.txt
  rm -rf $OUT/p$i
done
