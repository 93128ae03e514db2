"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch and
mean dispatch duration.   python tools/pmc_summary.py gpurun_out/pmc1/runc/*_counter_collection.csv ..."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
meta = {}
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("clm::", "").replace("void ", "").split("(")[0][:48]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[(n, r["Dispatch_Id"], f)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        meta[n] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
byk = defaultdict(list)
for (n, _, _), d in dur.items():
    byk[n].append(d)
for n in sorted(acc, key=lambda k: -sum(byk[k])):
    if sum(byk[n]) < 50:
        continue
    d = byk[n]
    print(f"\n{n}  dispatches={len(d)} avg_us={sum(d)/len(d):.1f}  wg={meta[n][0]} lds={meta[n][1]} vgpr={meta[n][2]} agpr={meta[n][3]} sgpr={meta[n][4]} scratch={meta[n][5]}")
    for c, v in sorted(acc[n].items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}")
