#!/bin/bash
# bench.py at several --chunk-reads values, interleaved twice (same-box comparison)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for c in "$@"; do
    timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --chunk-reads $c --no-cpu-baseline > $R/gpurun_out/sweep.log 2>&1 || { tail -3 $R/gpurun_out/sweep.log; exit 1; }
    python3 - "$c" <<PY
import json, sys
d = json.loads([l for l in open("$R/gpurun_out/sweep.log") if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 2), {k: round(v * d["ms_per_step"], 2) for k, v in d["stage_ms_share"].items()})
PY
  done
done
