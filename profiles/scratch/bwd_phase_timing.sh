#!/bin/bash
# Where the fused affine backward kernel's time goes: timing-only builds without single phases (NF_BWD_SKIP, see
# csrc/nf_affine2d_bwd.hip), training step at 2^20 rows.  Build here (CPU box), run on the GPU box:
#   bash profiles/scratch/bwd_phase_timing.sh build ; gpurun -- bash profiles/scratch/bwd_phase_timing.sh run
set -e
MASKS="${MASKS:-0 1 2 4 8 16 15 30 27}"
if [ "$1" = build ]; then
  for m in $MASKS; do
    rm -rf normalizing_flows_amd/csrc/build_skip$m; cp -r normalizing_flows_amd/csrc/build normalizing_flows_amd/csrc/build_skip$m
    rm -f normalizing_flows_amd/csrc/build_skip$m/nf_affine2d_bwd.o
    make -s -C normalizing_flows_amd/csrc EXTRA=-DNF_BWD_SKIP=$m BUILD=build_skip$m OUT=../libnf_hip_skip$m.so
  done
else
  for m in $MASKS; do
    echo "skip=$m"; NF_HIP_LIBRARY=$PWD/normalizing_flows_amd/libnf_hip_skip$m.so NF_TRAIN_LG=20 python3 profiles/bench_configs.py train | grep -o '"ms_stream": [0-9.]*'
  done
fi
