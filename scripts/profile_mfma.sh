#!/bin/bash
# MFMA-busy evidence for the batched coarse quantiser GEMM (PMC pass on its own, bench.py cut short)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_mfma
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --no-recall --kmeans-iters 1 > "$OUT/pmc.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --no-recall --kmeans-iters 1 > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
cnt=collections.defaultdict(list)
for f in glob.glob(out+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dist_gemm_kernel<false' in r['Kernel_Name']: cnt[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[]
for f in glob.glob(out+'/trace/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dist_gemm_kernel<false' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("dist_gemm_kernel<false_kernel: [1024 x 768] . [768 x 4096] f32, v_mfma_f32_32x32x2_f32")
for k,v in sorted(cnt.items()): print("  %-30s mean %.5g (n=%d)"%(k,sum(v)/len(v),len(v)))
if dur:
    d=sum(dur)/len(dur); fl=2*1024*4096*768
    print("  kernel time (trace) mean %.1f us min %.1f us -> %.1f TFLOP/s mean, %.1f TFLOP/s best = %.1f %% / %.1f %% of the 157.3 TFLOP/s f32 MFMA peak"%(d,min(dur),fl/d/1e6,fl/min(dur)/1e6,fl/d/1e6/157.3*100,fl/min(dur)/1e6/157.3*100))
PY
