#!/bin/bash
# Run GPU steps one after the other on the gpurun box: a step that FAILS (tests red) does not stop the following ones, a step that
# is KILLED by its timeout (rc 124 / 137: a hang) does -- no further GPU step is started after a hang.
#   tools/gpu_steps.sh OUTDIR "SECONDS|name|command" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  secs=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (timeout $secs s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping"; exit $rc; fi
done
