#!/bin/bash
# Build K4 tuning variants (tiles per workgroup x waves/SIMD bound) into build/abl/ for tools/k1_bench.py split.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/abl
for v in "2 3" "8 3" "4 4" "4 6" "4 8"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_K4_SUBS=$1 -DP25FE_K4_WPS=$2 -I$ROOT/include -shared \
     -o $ROOT/build/abl/libp25fe_k4_s$1w$2.so $ROOT/p25rx_amd/csrc/p25fe_api.hip &
done
wait
