#!/bin/bash
# Same-box A/B of library builds (build/variants/libp25fe_<tag>.so), interleaved, per-kernel split from the library's events.
# usage: bash tools/k1_ab.sh <rounds> <fmt> tag[:ENV=VAL,...] ...
R=$1; FMT=$2; shift 2
for r in $(seq 1 $R); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    out=$(env $envs P25FE_LIB=$PWD/build/variants/libp25fe_$tag.so python3 tools/k1_bench.py 600 40 split 1 $FMT 2>&1 | tail -2 | tr '\n' ' ')
    echo "$spec | $out"
  done
done
