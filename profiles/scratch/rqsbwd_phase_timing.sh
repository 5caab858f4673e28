#!/bin/bash
# Prices of the phases of rqs2d_bwd::chain_kernel: TIMING-ONLY builds (-DNF_RQSBWD_SKIP=mask: wrong results, same remaining
# instruction stream) of the spline flow's training step at 2^20 rows.  The variants are cross-compiled in the dev
# container by build_rqsbwd_skip.sh into profiles/scratch/_skip/ (git-ignored) and travel with the snapshot.
cd "$(dirname "$0")/../.."
echo "skip 0: $(NF_TRAINC_LG=20 python profiles/bench_configs.py trainc 2>&1 | grep -o '"ms_stream": [0-9.]*')"
for mask in 1 2 4 8 16 32 64 128 256; do
  echo "skip $mask: $(NF_HIP_LIBRARY=$PWD/profiles/scratch/_skip/libnf_skip_$mask.so NF_TRAINC_LG=20 python profiles/bench_configs.py trainc 2>&1 | grep -o '"ms_stream": [0-9.]*')"
done
echo "skip 0: $(NF_TRAINC_LG=20 python profiles/bench_configs.py trainc 2>&1 | grep -o '"ms_stream": [0-9.]*')"
