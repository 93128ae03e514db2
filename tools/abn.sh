#!/bin/bash
# Same-box comparison of N builds of the engine library:  bash tools/abn.sh ROUNDS libX.so libY.so ...   (bench args via ABN_ARGS)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for l in "$@"; do
    CLM_LIB=$R/chimeralm_amd/csrc/$l timeout -k 10 150 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline $ABN_ARGS > $R/gpurun_out/abn_run.log 2>&1 || { echo "run $l failed"; tail -3 $R/gpurun_out/abn_run.log; exit 1; }
    python3 - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/abn_run.log") if l.startswith("{")][-1])
print("$l", round(d["value"]), round(d["ms_per_step"],2), {k: round(x*d["ms_per_step"],2) for k,x in d["stage_ms_share"].items()})
PY
  done
done
