#!/bin/bash
# usage (GPU box): bash tools/mem_probe.sh [prof_target args]   L2 / memory-side view of the dominant kernel: three PMC passes
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/mem_probe
mkdir -p $out
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_target.py --iters 2 $EXTRA > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; exit 1; }
}
EXTRA="$*"
pass m1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
pass m2 TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum
pass m3 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum
pass m4 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
python3 - $out <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mfma" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-40s %.4e  (last of %d dispatches)" % (k, acc[k][-1], len(acc[k])))
PY
