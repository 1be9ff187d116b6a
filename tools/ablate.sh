#!/bin/bash
# Build measurement variants of the library with K1 truncated after stage N (P25FE_ABLATE=N) into build/abl/.
# N: 1 = load pipeline + LDS staging only, 2 = + decimator, 3 = + channel FIR, 4 = + FM, 5 = + boxcar sums (no output
# transpose / stores), 6 = everything on cache-resident loads, (full = the product).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/abl
for n in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_MEASURE -DP25FE_ABLATE=$n -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_abl$n.so $ROOT/p25rx_amd/csrc/p25fe_api.hip
done
# the same truncations with cache-resident window loads (the arithmetic side alone, stage by stage)
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_MEASURE -DP25FE_ABLATE=$n -DP25FE_ABLATE_CACHED -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_abl${n}c.so $ROOT/p25rx_amd/csrc/p25fe_api.hip
done
# K2 truncated after the sign-bit screen (1) / after the exact test of the screened positions (2)
for n in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_MEASURE -DP25FE_ABLATE_DET=$n -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_det$n.so $ROOT/p25rx_amd/csrc/p25fe_api.hip
done
