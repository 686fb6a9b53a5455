"""GPU busy time vs wall time from a rocprofv3 --kernel-trace CSV: how much of a training phase the GPU sits idle
between launches (host latency), and which kernels fill the rest.

    python tools/trace_gaps.py <..._kernel_trace.csv> [--skip 0.3]      (skips the first fraction: warm-up)
"""
import argparse
import csv
import collections

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--skip", type=float, default=0.3)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * a.skip):]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    per[k.split("(")[0][:70]][0] += e - s
    per[k.split("(")[0][:70]][1] += 1
print("launches %d, span %.3f ms, GPU busy (union of kernels) %.3f ms = %.1f %%, idle %.1f %%" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, 100.0 - 100.0 * busy / span))
for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %6.2f %% of span  %7d launches  %8.1f us avg  %s" % (100.0 * t / span, n, t / n / 1e3, k))
