#!/bin/bash
# Cross-compiles the TIMING-ONLY variants of linear_rqs_small_m_kernel (-DNF_SMR_SKIP=mask, one phase removed each) in the
# dev container into profiles/scratch/_skip/ (git-ignored; they travel to the GPU box with the snapshot).
# Needs an up-to-date normalizing_flows_amd/csrc/build/ (make).  Use: NF_HIP_LIBRARY=.../libnf_smr_skip_<mask>.so
set -e
cd "$(dirname "$0")/../../normalizing_flows_amd/csrc"
mkdir -p ../../profiles/scratch/_skip
for mask in ${@:-1 2 4 8 15}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. \
    -DNF_SMR_SKIP=$mask -c nf_small_m_rqs.hip -o /tmp/smr_skip_$mask.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/scratch/_skip/libnf_smr_skip_$mask.so \
    $(ls build/*.o | grep -v nf_small_m_rqs.o) /tmp/smr_skip_$mask.o -ldl -lpthread
done
echo built
