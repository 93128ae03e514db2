# final collection of round 4: the default and the 32k-bp profile rounds, the other precisions' bench lines (exact fp32 and fp16x3 on
# both nets, the mode's second level), BASELINE's C2 and the per-GPU shards of C4 / C5, the end-to-end run, a power log of the forward
set -o pipefail
O=gpurun_out/r04z; mkdir -p $O gpurun_out/r04 gpurun_out/r04_32k
bash tools/profile_round.sh r04 > $O/round.log 2>&1; tail -n 2 $O/round.log
cp profiles/r04_bench.json profiles/r04_kernel_stats.csv profiles/r04_pmc_summary.txt profiles/r04_traffic.json $O/
bash tools/profile_round.sh r04_32k --bases 32768 --batch 32 > $O/round_32k.log 2>&1; tail -n 2 $O/round_32k.log
cp profiles/r04_32k_bench.json profiles/r04_32k_kernel_stats.csv profiles/r04_32k_pmc_summary.txt profiles/r04_32k_traffic.json $O/
B="timeout -k 10 250 python bench.py --no-cpu-baseline"
$B --precision fp32 --steps 4 --warmup 1 --no-guard > $O/r04_fp32_bench.json 2>/dev/null; cut -c1-160 $O/r04_fp32_bench.json
$B --precision fp16x3 --steps 8 --warmup 2 > $O/r04_fp16x3_bench.json 2>/dev/null; cut -c1-160 $O/r04_fp16x3_bench.json
$B --mlp-lo --no-fp32-leg > $O/r04_bench_level2.json 2>/dev/null; cut -c1-160 $O/r04_bench_level2.json
$B --bases 4096 --batch 64 --steps 20 --warmup 5 --no-fp32-leg > $O/r04_c2_4k_b64_fp16c.json 2>/dev/null; cut -c1-160 $O/r04_c2_4k_b64_fp16c.json
$B --batch 32 --steps 30 --warmup 5 --no-fp32-leg > $O/r04_c4_shard_8k_b32_fp16c.json 2>/dev/null; cut -c1-160 $O/r04_c4_shard_8k_b32_fp16c.json
$B --bases 32768 --batch 4 --steps 10 --warmup 2 --no-fp32-leg > $O/r04_c5_shard_32k_b4_fp16c.json 2>/dev/null; cut -c1-160 $O/r04_c5_shard_32k_b4_fp16c.json
$B --net transformer --precision fp32 --steps 3 --warmup 1 --no-fp32-leg > $O/r04_tf_fp32_bench.json 2>/dev/null; cut -c1-160 $O/r04_tf_fp32_bench.json
$B --net transformer --precision fp16x3 --steps 6 --warmup 2 --no-fp32-leg > $O/r04_tf_fp16x3_bench.json 2>/dev/null; cut -c1-160 $O/r04_tf_fp16x3_bench.json
$B --net transformer --precision fp16c --steps 10 --warmup 3 --no-fp32-leg > $O/r04_tf_bench.json 2>/dev/null; cut -c1-160 $O/r04_tf_bench.json
timeout -k 10 300 python tools/e2e_bench.py > $O/r04_e2e.txt 2>&1; tail -n 3 $O/r04_e2e.txt
timeout -k 10 300 python tools/power_log.py all --seconds 5 --csv $O/r04_power.csv > $O/r04_power.txt 2>&1; grep -E "==|power_w|sclk" $O/r04_power.txt | head -20
