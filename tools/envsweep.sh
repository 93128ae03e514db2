#!/bin/bash
# Same-box comparison of bench.py under several "ENV=val,--flag val" settings:  bash tools/envsweep.sh ROUNDS "CLM_X=1 --chunk-reads 64" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for cfg in "$@"; do
    envs=(); args=()
    for w in $cfg; do if [[ $w == *=* && $w != --* ]]; then envs+=("$w"); else args+=("$w"); fi; done
    env "${envs[@]}" timeout -k 10 150 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "${args[@]}" > $R/gpurun_out/sweep_run.log 2>&1 || { echo "run [$cfg] failed"; tail -3 $R/gpurun_out/sweep_run.log; exit 1; }
    python3 - "$cfg" <<PY
import json, sys
d=json.loads([l for l in open("$R/gpurun_out/sweep_run.log") if l.startswith("{")][-1])
print(sys.argv[1].ljust(40), round(d["value"]), round(d["ms_per_step"],2), {k: round(x*d["ms_per_step"],2) for k,x in d["stage_ms_share"].items()})
PY
  done
done
