#!/bin/bash
# Same-box A/B of K6 / K0 builds (build/variants/libp25fe_<tag>.so; tag "cur" = the in-tree library), interleaved.
# usage: bash tools/k6_ab.sh <rounds> <chz|k0> tag ...
R=$1; MODE=$2; shift 2
for r in $(seq 1 $R); do
  for tag in "$@"; do
    lib=$PWD/build/variants/libp25fe_$tag.so; [ "$tag" == "cur" ] && lib=$PWD/p25rx_amd/libp25fe.so
    out=$(P25FE_LIB=$lib python3 tools/k1_bench.py 600 12 $MODE 1 cf32 2>&1 | tail -1)
    echo "$tag | $out"
  done
done
