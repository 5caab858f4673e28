# launch geometry of the variant A forward kernel (development knobs NF_SPLIT2D_SAMPLES / NF_SPLIT2D_GRID), 2^24 rows
for s in ${SWEEP_S:-2 4 8}; do for g in ${SWEEP_G:-2048 4096 8192 16384}; do echo "S=$s G=$g"; NF_BENCH_A_LG=${SWEEP_LG:-24} NF_SPLIT2D_SAMPLES=$s NF_SPLIT2D_GRID=$g python profiles/bench_configs.py a 2>&1 | tail -1; done; done
