#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_all.sh <tag>
# For each of the five BASELINE configs: bench.py line + rocprofv3 kernel-trace + the five PMC passes of the fused
# loss+grad kernel (tools/profile.sh).  Results under gpurun_out/<tag>/<workload>/ ; copy summaries to profiles/.
set -u
tag=$1
mkdir -p gpurun_out/$tag
for w in burgers burgers_9x20 kdv5 ch gnss kdv7; do
  echo "== $w"
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err || { echo "bench $w failed"; exit 1; }
  BENCH_JSON=gpurun_out/$tag/bench_$w.json bash tools/profile.sh $tag/$w --workload $w > gpurun_out/$tag/profile_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/$tag/profile_$w.log; exit 1; }
  grep "bench.py kernel_ms" gpurun_out/$tag/$w/summary.txt | cut -c1-200
done
# asymptotic fraction of the K = 5 and K = 2 instantiations: the same kernels at 1M points (BASELINE quotes 100k / 10k)
for w in kdv5 burgers; do
  echo "== $w at 1M points"
  bash tools/profile.sh $tag/${w}_1M --workload $w --n 1000000 > gpurun_out/$tag/profile_${w}_1M.log 2>&1 || { echo "profile $w 1M failed"; exit 1; }
  grep -m3 -A3 "kernel trace" gpurun_out/$tag/${w}_1M/summary.txt | cut -c1-200
done
