#!/bin/bash
# kernel trace of the timed steps only: per-kernel mean durations of one cfg3 search step (build kernels filtered out)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_step
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --no-recall --kmeans-iters 1 "$@" > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
rows=[]
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_trace.csv',recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the steps = everything from the first prescan/scan IvfSrc kernel's preceding stage kernel; simply take dispatches whose name is in the step set
agg=collections.OrderedDict()
step_names=('stage_queries','dist_gemm_kernel<false','coarse_select','plan_kernel','group_kernel','scatter_pairs','items_kernel','gather_qblocks','prescan_kernel','scan_kernel<16','scan_kernel<8, 0, vers::IvfSrc','ivf_rescore','fallback_','ivf_merge','fillBuffer','rank_merge','copyBuffer')
first=None
for i,r in enumerate(rows):
    if 'prescan_kernel' in r['Kernel_Name'] or 'IvfSrc' in r['Kernel_Name']:
        first=i; break
# walk back to the stage kernel that starts this step
j=first
while j>0 and 'stage_queries' not in rows[j]['Kernel_Name']: j-=1
sel=rows[j:]
n_steps=sum(1 for r in sel if 'prescan_kernel' in r['Kernel_Name'] or ('scan_kernel' in r['Kernel_Name'] and 'IvfSrc' in r['Kernel_Name']))
tot=0
for r in sel:
    k=r['Kernel_Name'].replace('vers::','')[:60]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d; tot+=d
span=(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/1e3
print(f"{n_steps} steps; kernel time per step {tot/n_steps:.1f} us; wall span per step {span/n_steps:.1f} us")
for k,a in agg.items(): print(f"  {a[0]/n_steps:5.1f} x {a[1]/a[0]:9.1f} us = {a[1]/n_steps:9.1f} us/step  {k}")
PY
