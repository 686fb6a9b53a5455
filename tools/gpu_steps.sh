#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_steps.sh <outdir> <<'STEPS' ... one "name|seconds|command" per line
# Runs the steps in order, each under its own time limit, output to gpurun_out/<outdir>/<name>.log.  A step that ends
# with a plain failure (exit 1: failed assertions) does not stop the list; anything else (time limit, abort,
# segmentation fault, GPU fault) does: nothing more is started on the GPU after that.
set -u
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  echo "== $name ($(date +%T))"
  timeout -k 10 $secs bash -c "$cmd" > $out/$name.log 2>&1
  rc=$?
  echo "   rc=$rc"; tail -3 $out/$name.log | cut -c1-300
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name (rc $rc)"; exit $rc; fi
done
exit 0
