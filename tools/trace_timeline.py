"""Timeline of one replayed closure out of a rocprofv3 --kernel-trace CSV: tools/trace_timeline.py <kernel_trace.csv> <substring of the anchor kernel>
prints the launches around the middle occurrence of the anchor kernel (start, end, duration in us, queue)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i = mid[len(mid) // 2]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[max(0, i - 8):i + 45]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f q%s grid=%-7s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Grid_Size_X"], r["Kernel_Name"][:64]))
