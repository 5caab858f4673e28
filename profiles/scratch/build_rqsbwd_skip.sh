#!/bin/bash
# Cross-compiles the TIMING-ONLY variants of rqs2d_bwd::chain_kernel (-DNF_RQSBWD_SKIP=mask, one phase removed each) in the
# dev container into profiles/scratch/_skip/ (git-ignored; they travel to the GPU box with the snapshot), for
# rqsbwd_phase_timing.sh.  Needs an up-to-date normalizing_flows_amd/csrc/build/ (make).
set -e
cd "$(dirname "$0")/../../normalizing_flows_amd/csrc"
mkdir -p ../../profiles/scratch/_skip
for mask in 1 2 4 8 16 32 64 128 256; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -fno-honor-nans -fno-slp-vectorize \
    -DNF_RQSBWD_SKIP=$mask -c nf_rqs2d_bwd.hip -o /tmp/rqsbwd_skip_$mask.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/scratch/_skip/libnf_skip_$mask.so \
    $(ls build/*.o | grep -v nf_rqs2d_bwd.o) /tmp/rqsbwd_skip_$mask.o -ldl -lpthread
done
echo built
