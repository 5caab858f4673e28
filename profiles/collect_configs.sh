#!/bin/bash
# rocprofv3 summaries of the other BASELINE configs (run on the GPU box through gpurun):
#   configs[2] (2-D spline flow, fused kernel, both matmul modes) and configs[4] (MNIST shape, general path).
# Kernel trace + stats, then separate --pmc passes; summaries land in profiles/r01/config_<c|d>_*.
set -e
export TMPDIR=/tmp
for cfg in ${@:-c d}; do
  OUT=gpurun_out/prof_cfg_$cfg
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 profiles/bench_configs.py $cfg > $OUT/kt.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/sq1 -- python3 profiles/bench_configs.py $cfg > $OUT/sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES -f csv -d $OUT/sq2 -- python3 profiles/bench_configs.py $cfg > $OUT/sq2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- python3 profiles/bench_configs.py $cfg > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- python3 profiles/bench_configs.py $cfg > $OUT/write.log 2>&1
  python3 profiles/summarize.py $OUT r01/config_$cfg
done
echo collected
