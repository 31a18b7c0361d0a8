"""Per-query timeline of a single-query kernel trace (scripts/trace_b1.sh): mean duration of every kernel of the query's
chain and the mean gap in front of it, over the warm queries.  usage: timeline_b1.py <trace dir> [anchor substring]"""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'scan1_kernel'
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
idx = idx[len(idx) // 2:]  # the warm half
per = idx[1] - idx[0]
dur = defaultdict(list); gap = defaultdict(list); order = []
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != per: continue
    for j in range(a, b):
        r = rows[j]; name = r['Kernel_Name'].replace('vers::', '')[:60]
        key = (j - a, name)
        if key not in order: order.append(key)
        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        gap[key].append((int(r['Start_Timestamp']) - int(rows[j - 1]['End_Timestamp'])) / 1e3)
tot = 0.0
print(f"{'gap_us':>8} {'dur_us':>8}  kernel   ({len(idx)-1} queries, {per} kernels each)")
for key in sorted(order):
    g = sum(gap[key]) / len(gap[key]); d = sum(dur[key]) / len(dur[key]); tot += g + d
    print(f"{g:8.2f} {d:8.2f}  {key[1]}")
print(f"per query: {tot:.1f} us")
