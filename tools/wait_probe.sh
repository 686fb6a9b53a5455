#!/bin/bash
# usage (GPU box): bash tools/wait_probe.sh [prof_target args]   where the waves of the dominant kernel wait: two PMC passes
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/wait_probe
mkdir -p $out
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_target.py --iters 2 $EXTRA > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; exit 1; }
}
EXTRA="$*"
pass w1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass w2 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
python3 - $out <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mfma" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-28s %.4e  (last of %d dispatches)" % (k, acc[k][-1], len(acc[k])))
PY
