"""Summarise rocprofv3 --pmc csv passes: per kernel, counter value per dispatch (mean)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        rows = list(csv.DictReader(f))
    per = defaultdict(float)
    for r in rows:
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for fn in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats ==")
    with open(fn) as f:
        for i, line in enumerate(f):
            if i < 4:
                print(line.strip()[:160])
for k in acc:
    if "pinn" not in k:
        continue
    print("==", k[:90])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("  %-32s %.6e  (dispatches %d)" % (c, sum(v) / len(v), len(v)))
