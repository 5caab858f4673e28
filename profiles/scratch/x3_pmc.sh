#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/x3pmc
mkdir -p $OUT
export MODE=bf16x3
export NF_AFFINE2D_X3_TUNE=${1:-4,8}
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -f csv -d $OUT/sq1 -- python3 profiles/x3_tune.py one > $OUT/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -f csv -d $OUT/sq2 -- python3 profiles/x3_tune.py one > $OUT/sq2.log 2>&1 &&
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_EXP_GDS -f csv -d $OUT/sq3 -- python3 profiles/x3_tune.py one > $OUT/sq3.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ["sq1","sq2","sq3"]:
    for f in glob.glob(f"gpurun_out/x3pmc/{d}/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'chain_kernel' in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c,v in sorted(agg.items()):
            print(f"{c:32s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
PY
