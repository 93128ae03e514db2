# second collection of round 3 (after the 256-read chunks): 32k profile, shard lines, power log
set -o pipefail
O=gpurun_out/r03y; mkdir -p $O
bash tools/profile_round.sh r03_32k --bases 32768 --batch 32 --steps 6 > $O/round_32k.log 2>&1; tail -n 2 $O/round_32k.log
timeout -k 10 200 python bench.py --bases 32768 --batch 4 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-leg > $O/c5_shard.json 2> $O/c5_shard.err; cut -c1-200 $O/c5_shard.json
timeout -k 10 200 python bench.py --bases 4096 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/c2_fp16c.json 2>/dev/null; cut -c1-160 $O/c2_fp16c.json
timeout -k 10 200 python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/c4_shard.json 2>/dev/null; cut -c1-160 $O/c4_shard.json
timeout -k 10 300 python tools/power_log.py all --seconds 5 --csv $O/power.csv > $O/power.txt 2>&1; grep -E "telemetry|==|power_w|sclk" $O/power.txt | head -30
