#!/bin/bash
# Build K6 (channeliser) tuning variants: c1 group size x waves/SIMD bound, into build/abl/ for tools/k1_bench.py chz.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build/abl
for v in "1 4" "1 5" "2 3" "3 3" "4 3" "6 2"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
     -fhip-fp32-correctly-rounded-divide-sqrt -DP25FE_CZ_G=$1 -DP25FE_CZ_WPS=$2 -I$ROOT/include -shared \
     -Rpass-analysis=kernel-resource-usage -o $ROOT/build/abl/libp25fe_cz_g$1w$2.so $ROOT/p25rx_amd/csrc/p25fe_api.hip 2> $ROOT/build/abl/cz_g$1w$2.log &
done
wait
for f in $ROOT/build/abl/cz_*.log; do echo $(basename $f) $(grep -A9 "k_channelise" $f | grep -E "VGPRs:|Scratch" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '); done
