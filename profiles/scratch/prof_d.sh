#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_d
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 profiles/bench_configs.py d > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/sq1 -- python3 profiles/bench_configs.py d > $OUT/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES -f csv -d $OUT/sq2 -- python3 profiles/bench_configs.py d > $OUT/sq2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/prof_d/kt/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
for d in ["sq1","sq2"]:
    for f in glob.glob(f"gpurun_out/prof_d/{d}/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if 'gemm' in k:
                agg[(k[-40:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k,c),v in sorted(agg.items()):
            print(f"{k:42s} {c:28s} n={len(v):4d} max={max(v):16.1f} mean={sum(v)/len(v):14.1f}")
PY
