#!/bin/bash
# run build_dev/lib_<name>.so (all, or the names given) through tools/prof_target.py; one line per variant
# WORKLOAD=gnss tools/run_variants.sh ...   selects the bench workload
cd "$(dirname "$0")/.."
names="$@"
[ -z "$names" ] && names=$(ls build_dev/lib_*.so | sed 's/.*lib_//; s/\.so//')
for n in $names; do
  r=$(timeout -k 10 120 python3 tools/prof_target.py --iters 8 --workload ${WORKLOAD:-kdv7} ${CHECK:+--check} --lib build_dev/lib_$n.so 2>&1 | grep -v "^iter [0-4]" | tr "\n" " ") || { echo "$n: FAILED $r"; exit 1; }
  echo "$n: $r"
done
