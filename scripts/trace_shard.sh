#!/bin/bash
# kernel trace of one rank's steps of a W-way sharded search (scripts/emulate_shard.py): per-kernel mean per step.
# usage: scripts/trace_shard.sh W   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_shard
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/emulate_shard.py" "${1:-8}" 0 > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
rows=[]
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_trace.csv',recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'prescan_kernel' in r['Kernel_Name']]
# the last 20 steps: from the stage kernel before the 20th-from-last scan
j=idx[-20]
while j>0 and 'stage' not in rows[j]['Kernel_Name']: j-=1
sel=rows[j:]
agg=collections.OrderedDict(); tot=0
for r in sel:
    k=r['Kernel_Name'].replace('vers::','')[:70]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d; tot+=d
span=(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/1e3
print(f"20 steps; kernel time per step {tot/20:.1f} us; wall span per step {span/20:.1f} us")
for k,(c,t) in agg.items(): print(f"{t/20:9.1f} us/step  x{c/20:.1f}  {k}")
PY
