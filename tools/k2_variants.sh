#!/bin/bash
# Build measurement variants of K2 truncated after stage N (P25FE_ABLATE2=N: 1 staging, 2 + correlation, 3 + peak pick)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/abl
for n in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_ABLATE2=$n -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_k2a$n.so $ROOT/p25rx_amd/csrc/p25fe_api.hip &
done
wait
