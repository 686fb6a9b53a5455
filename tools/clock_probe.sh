#!/bin/bash
# usage (GPU box): bash tools/clock_probe.sh <lib> ...   shader clock of the dominant kernel under its own load:
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration, per launch, for development libraries
set -u
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  out=$PWD/gpurun_out/clock_$name
  mkdir -p $out
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out -- python3 tools/prof_target.py --iters 6 --lib $lib > $out/log.txt 2>&1 || { echo "$name: failed"; tail -3 $out/log.txt; exit 1; }
  python3 - "$out" "$name" <<'PY'
import csv, glob, sys
out, name = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mfma" not in r["Kernel_Name"]: continue
        d = rows.setdefault(r["Dispatch_Id"], {})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        for k in ("Start_Timestamp", "End_Timestamp"):
            if k in r: d[k] = float(r[k])
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mfma" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(rows, key=int):
    d = rows[k]; t = dur.get(k) or (d.get("End_Timestamp", 0) - d.get("Start_Timestamp", 0))
    print("%s dispatch %s: %.3f ms, GRBM_GUI_ACTIVE %.4e -> %.3f GHz; wave cycles %.3e, wait_inst %.3e, mfma busy %.3e" % (
        name, k, t * 1e-6, d.get("GRBM_GUI_ACTIVE", 0), d.get("GRBM_GUI_ACTIVE", 0) / 8 / t if t else 0,
        d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_INST_ANY", 0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
PY
done
