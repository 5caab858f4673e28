#!/bin/bash
# Round-1 profile collection for the headline bench (run on the GPU box through gpurun):
#   kernel trace + stats, then SEPARATE --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ sets),
#   then the FETCH_SIZE calibration stream.  Outputs under gpurun_out/prof_r01/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01
mkdir -p $OUT
B="--no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 bench.py --steps 50 --warmup 10 $B > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/sq1 -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES -f csv -d $OUT/sq2 -- python3 bench.py --steps 5 --warmup 2 $B > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal -- python3 profiles/calibrate_fetch.py > $OUT/cal.log 2>&1
echo collected
