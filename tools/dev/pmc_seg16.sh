cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_seg16; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  CLM_SEG16=1 timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o seg16 -- python3 $R/bench.py --bases 32768 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg > $O/$c.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "seg" in k: print(c, k, len(v), sum(v)/len(v))
PY
