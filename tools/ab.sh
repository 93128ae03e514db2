#!/bin/bash
# Same-box A/B of two builds of the engine library:  bash tools/ab.sh libclm_A.so libclm_B.so [bench args...]
# (alternating runs; box-to-box variation of the headline number is +-3 %, larger than most single optimisations)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=$R/chimeralm_amd/csrc/$1; B=$R/chimeralm_amd/csrc/$2; shift 2
for i in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    CLM_LIB=$lib timeout -k 10 150 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/ab_$v.log 2>&1 || { echo "run $v failed"; tail -3 $R/gpurun_out/ab_$v.log; exit 1; }
    python3 - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/ab_$v.log") if l.startswith("{")][-1])
print("$v", round(d["value"]), round(d["ms_per_step"],2), {k: round(x*d["ms_per_step"],2) for k,x in d["stage_ms_share"].items()})
PY
  done
done
