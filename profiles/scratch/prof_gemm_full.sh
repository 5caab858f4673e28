#!/bin/bash
# A/B of the two GEMM main loops on configs[4] (bench_configs.py d): kernel trace + two SQ counter sets each.
export TMPDIR=/tmp
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"
for mode in full nofull; do
  if [ $mode = nofull ]; then export NF_GEMM_NO_FULL=1; else unset NF_GEMM_NO_FULL; fi
  OUT=gpurun_out/prof_gemm_$mode
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 profiles/bench_configs.py d > $OUT/kt.log 2>&1
  rocprofv3 --pmc $SQ1 -f csv -d $OUT/sq1 -- python3 profiles/bench_configs.py d > $OUT/sq1.log 2>&1
  rocprofv3 --pmc $SQ2 -f csv -d $OUT/sq2 -- python3 profiles/bench_configs.py d > $OUT/sq2.log 2>&1
  python3 - $OUT <<'PY'
import csv,glob,sys,collections
src=sys.argv[1]
print(src)
for r in csv.DictReader(open(glob.glob(f"{src}/kt/*/*_kernel_stats.csv")[0])):
    if 'gemm' in r['Name']: print("  ", r['Name'][:60], r['Calls'], r['AverageNs'])
for d in ("sq1","sq2"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{src}/{d}/*/*_counter_collection.csv")[0])):
        if 'gemm_rqs_kernel' in r['Kernel_Name'] or 'gemm_fast_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print("  ", k, sum(v)/len(v))
PY
done
