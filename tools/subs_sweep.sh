#!/bin/bash
# K1 segment-length sweep on one box (P25FE_SUBS: sub-tiles per workgroup), interleaved.  usage: bash tools/subs_sweep.sh <rounds> <fmt> subs...
R=$1; FMT=$2; shift 2
for r in $(seq 1 $R); do
  for s in "$@"; do
    out=$(P25FE_SUBS=$s python3 tools/k1_bench.py 600 40 split 1 $FMT 2>&1 | tail -1)
    echo "subs=$s | $out"
  done
done
