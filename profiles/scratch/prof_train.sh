#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_train
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 profiles/bench_configs.py train > $OUT/kt.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_train/kt/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MaxNs"], r["Percentage"])
PY
