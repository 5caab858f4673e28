#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_x3
mkdir -p $OUT
B="--no-cpu-baseline --hidden-matmul bf16x3"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 bench.py --steps 50 --warmup 10 $B > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/write.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/sq1 -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES -f csv -d $OUT/sq2 -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/sq2.log 2>&1 &&
python3 profiles/summarize.py $OUT r01/bf16x3_bench_n1
cat $OUT/kt.log | tail -1 | cut -c1-600
