#!/bin/bash
# Round-4 profile collection (run on the GPU box through gpurun; ~6 minutes):
#   headline bench:  kernel trace + stats, then SEPARATE --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ sets)
#   configs[2], [4]: the same on profiles/bench_configs.py c / d;  a: the src/real_nvp.py flow (variant A)
# rocprofv3 gets the program itself after `--` (python3 ...), never a wrapper.  Summaries land in profiles/r04/
# (profiles/summarize.py): <tag>_kernel_stats.csv = the --stats table, <tag>_pmc_summary.json = per-kernel counter
# means with hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B; reads doubled as MI355X_MICROARCH.md's HBM
# section prescribes for gfx950).
set -e
export TMPDIR=/tmp
mkdir -p profiles/r04
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES"
collect() {  # collect <outdir> <tag> <program args...>   (KT_ARGS, if set, replaces the arguments of the kernel-trace pass)
  local OUT=$1 TAG=$2; shift 2
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- python3 ${KT_ARGS:-"$@"} > $OUT/kt.log 2>&1
  rocprofv3 --pmc $SQ1 -f csv -d $OUT/sq1 -- python3 "$@" > $OUT/sq1.log 2>&1
  rocprofv3 --pmc $SQ2 -f csv -d $OUT/sq2 -- python3 "$@" > $OUT/sq2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- python3 "$@" > $OUT/write.log 2>&1
  python3 profiles/summarize.py $OUT $TAG
}
for what in ${@:-bench c d a}; do
  case $what in
    # kernel trace: the driver's own command (all four configs, blocks repeated to 0.5 s) minus the CPU baseline;
    # counters: the headline alone, one block
    bench) KT_ARGS="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" \
           collect gpurun_out/prof_r04_bench r04/final_bench_n1 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-variants --no-other-configs ;;
    c) collect gpurun_out/prof_r04_c r04/config_c profiles/bench_configs.py c ;;
    d) collect gpurun_out/prof_r04_d r04/config_d profiles/bench_configs.py d ;;
    a) collect gpurun_out/prof_r04_a r04/config_a profiles/bench_configs.py a ;;
    # training steps at 2^20 rows: the fused affine step, the spline flow's general step
    train) NF_TRAIN_LG=20 collect gpurun_out/prof_r04_train r04/train_b_2p20 profiles/bench_configs.py train ;;
    trainc) NF_TRAINC_LG=20 collect gpurun_out/prof_r04_trainc r04/train_c_2p20 profiles/bench_configs.py trainc ;;
    # the reference's own batch (128 rows of the 784-D flow): log_prob call and training step
    mnist128) NF_MNIST_ROWS=128 collect gpurun_out/prof_r04_mnist128 r04/mnist_b128_logprob profiles/mnist_small_batch.py ;;
    mnist128t) collect gpurun_out/prof_r04_mnist128t r04/train_mnist_b128 profiles/mnist_train_step.py ;;
  esac
done
# the box keeps nothing: the summaries travel back under gpurun_out/ and are copied into profiles/r04/ from there
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04/* gpurun_out/profiles_r04/
echo collected
