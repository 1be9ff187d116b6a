#!/bin/bash
# Same-box A/B of K0 builds (build/variants/libp25fe_<tag>.so; tag "cur" = the in-tree library), interleaved.
# usage: bash tools/k0_ab.sh <rounds> tag ...
R=$1; shift
for r in $(seq 1 $R); do
  for tag in "$@"; do
    lib=$PWD/build/variants/libp25fe_$tag.so; [ "$tag" == "cur" ] && lib=$PWD/p25rx_amd/libp25fe.so
    out=$(P25FE_LIB=$lib python3 tools/k1_bench.py 600 40 k0 1 cf32 2>&1 | tail -1)
    echo "$tag | $out"
  done
done
