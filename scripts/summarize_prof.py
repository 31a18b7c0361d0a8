"""Condenses a scripts/profile_bench.sh output directory into a small text summary
(per-kernel time from the kernel trace; per-kernel PMC sums/means)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("vers::", "")
    return name if len(name) < 90 else name[:87] + "..."


# ---- kernel trace ----
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r["Kernel_Name"]]
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values()) or 1.0
print("== kernel trace (us) ==")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6}  kernel")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.1f} {a[2]:10.1f} {a[3]:10.1f} {100*a[1]/tot:6.2f}  {short(k)}")

# ---- PMC passes ----
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    cnt = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = cnt[r["Kernel_Name"]][r["Counter_Name"]]
            c[0] += 1; c[1] += float(r["Counter_Value"])
    print(f"\n== {os.path.basename(d)}: per-dispatch MEAN of each counter ==")
    for k, cs in sorted(cnt.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))[:6]:
        print("  " + short(k))
        for name, (n, s) in sorted(cs.items()):
            print(f"      {name:28s} mean {s/n:18.1f}   dispatches {n}")
