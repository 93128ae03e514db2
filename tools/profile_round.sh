#!/bin/bash
# Collect the evidence a round's numbers come from, on the GPU box:   bash tools/profile_round.sh r01
#   1. the default bench line (batch 256, 8k-bp, with the CPU baseline)
#   2. rocprofv3 --output-format csv --kernel-trace --stats of the same bench command
#   3. PMC passes (separate runs, counters only): HBM fetch, HBM write, SQ wave/wait/active cycles, MFMA busy
# Everything lands in gpurun_out/<tag>/; tools/profile_digest.py turns it into the small files kept under profiles/.
set -eo pipefail
TAG=${1:-r01}
shift || true
EXTRA="$*"            # extra bench.py arguments, e.g.  bash tools/profile_round.sh r01_tf --net transformer
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py $EXTRA"

timeout -k 10 300 $BENCH > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "[1/7] bench done"; cut -c1-200 "$O/bench_default.json"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$O/kt" -o "$TAG" -- $BENCH --no-cpu-baseline --no-fp32-leg > "$O/kt.log" 2>&1
echo "[2/7] kernel trace done"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$O/pmc_fetch" -o "$TAG" -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > "$O/pmc_fetch.log" 2>&1
echo "[3/7] FETCH_SIZE done"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$O/pmc_write" -o "$TAG" -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > "$O/pmc_write.log" 2>&1
echo "[4/7] WRITE_SIZE done"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d "$O/pmc_sq" -o "$TAG" -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > "$O/pmc_sq.log" 2>&1
echo "[5/7] SQ counters done"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d "$O/pmc_inst" -o "$TAG" -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > "$O/pmc_inst.log" 2>&1 || echo "inst counters pass failed (non-fatal)"
echo "[6/7] instruction counters done"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU -d "$O/pmc_icache" -o "$TAG" -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > "$O/pmc_icache.log" 2>&1 || echo "icache counters pass failed (non-fatal)"
echo "[7/7] instruction-cache counters done"
cd "$R" && python3 tools/profile_digest.py "$TAG" || echo "digest failed"
# the raw traces are large; keep the stats and counter csv files only
find "$O" -name '*_kernel_trace.csv' -size +8M -delete || true
