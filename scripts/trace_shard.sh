#!/bin/bash
# kernel trace of one rank's step of an N-way sharded search (scripts/emulate_shard.py N 0): per-kernel us per step (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-8}
OUT=$ROOT/gpurun_out/prof_shard$N
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/emulate_shard.py" $N 0 > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
python3 - "$OUT/trace" <<'PY'
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'prescan_kernel' in r['Kernel_Name']]
lo, hi = idx[-21], idx[-1]
agg = defaultdict(float)
for r in rows[lo:hi]: agg[r['Kernel_Name'].replace('vers::', '')[:60]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 / 20
span = (int(rows[hi]['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3 / 20
print(f"kernel time per step {sum(agg.values()):.1f} us; wall span per step {span:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print(f"  {v:8.1f} us/step  {k}")
PY
