#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_all.sh <tag>
# For each of the five BASELINE configs: bench.py line + rocprofv3 kernel-trace + the five PMC passes of the fused
# loss+grad kernel (tools/profile.sh).  Results under gpurun_out/<tag>/<workload>/ ; copy summaries to profiles/.
set -u
tag=$1
mkdir -p gpurun_out/$tag
for w in burgers kdv5 ch gnss kdv7; do
  echo "== $w"
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err || { echo "bench $w failed"; exit 1; }
  bash tools/profile.sh $tag/$w --workload $w > gpurun_out/$tag/profile_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/$tag/profile_$w.log; exit 1; }
  grep -A2 "kernel stats" gpurun_out/$tag/$w/summary.txt | tail -1 | cut -c1-150
done
