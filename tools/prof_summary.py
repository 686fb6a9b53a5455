"""Summary of one tools/profile.sh run (gpurun_out/<tag>/): the kernel trace with COLD launches apart, and the PMC passes.

    python3 tools/prof_summary.py gpurun_out/<tag> [bench.json]

Per kernel of the library: number of launches, the first `COLD` launches (code-object load, clocks ramping)
reported separately, and min / average / max of the warm ones - the figure bench.py's `roofline.kernel_ms` has to
agree with (a three-launch average that includes the cold one is not evidence).  Then, per kernel,
every counter of the separate `--pmc` passes (mean per dispatch).  With a bench.py JSON line as second argument the
bench's HIP-event kernel time is printed beside the profile's warm average.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

COLD = 3
root = sys.argv[1]
rows = []
for fn in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(fn) as f:
        rows += list(csv.DictReader(f))
by = defaultdict(list)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)     # microseconds
print("== kernel trace (us): launches | first %d (cold) | warm min / avg / max" % COLD)
main_avg = None
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if "pinn" not in k and "k_" not in k:
        continue
    cold, warm = v[:COLD], v[COLD:] or v
    avg = sum(warm) / len(warm)
    if main_avg is None:
        main_avg = avg
    print("%-90s %4d | %s | %.1f / %.1f / %.1f" % (k[:90], len(v), " ".join("%.1f" % c for c in cold), min(warm), avg, max(warm)))
if len(sys.argv) > 2 and main_avg is not None:
    line = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    kms = line["roofline"]["kernel_ms"]
    print("bench.py kernel_ms (HIP events, same build): %.4f ms; profile warm average of the dominant kernel: %.4f ms; "
          "difference %+.1f %%" % (kms, main_avg * 1e-3, 100.0 * (main_avg * 1e-3 - kms) / kms))
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        per = defaultdict(float)
        for r in csv.DictReader(f):
            per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for k in acc:
    if "pinn" not in k:
        continue
    print("==", k[:100])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("  %-32s %.6e  (dispatches %d)" % (c, sum(v) / len(v), len(v)))
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        print("  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction) = %.4e" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_VALU_FMA_F32" in c and "SQ_INSTS_MFMA" in c:
        nonm = c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]
        print("  non-MFMA VALU instructions %.4e, of which FMA %.1f %%" % (nonm, 100.0 * c["SQ_INSTS_VALU_FMA_F32"] / nonm))
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
        print("  LDS bank-conflict cycles / LDS active cycles = %.1f %%" % (100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
