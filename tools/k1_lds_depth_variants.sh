#!/bin/bash
# Measurement builds of the library with different LDS read-ahead depths in K1's FIR loops (P25FE_K1_LDS_DEPTH;
# 0 = compiler-scheduled volatile reads) into build/abl/.  Select one with P25FE_LIB=build/abl/libp25fe_ld<N>.so.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_K1_LDS_DEPTH=$n -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_ld$n.so $ROOT/p25rx_amd/csrc/p25fe_api.hip &
done
wait
