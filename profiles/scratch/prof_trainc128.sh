#!/bin/bash
export TMPDIR=/tmp
export NF_TRAINC_LG=7
OUT=gpurun_out/prof_trainc128
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 profiles/bench_configs.py trainc > $OUT/kt.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_trainc128/kt/*/*_kernel_stats.csv")[0]
tot=0
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
t=glob.glob("gpurun_out/prof_trainc128/kt/*/*_kernel_trace.csv")[0]
rows=list(csv.DictReader(open(t)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: take the final 140 kernels, print gaps
last=rows[-150:]
gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(last,last[1:])]
durs=[int(a["End_Timestamp"])-int(a["Start_Timestamp"]) for a in last]
print("last 150 kernels: sum dur %.1f us, sum gaps %.1f us, mean gap %.2f us, max gap %.1f us"%(sum(durs)/1e3,sum(gaps)/1e3,sum(gaps)/len(gaps)/1e3,max(gaps)/1e3))
PY
