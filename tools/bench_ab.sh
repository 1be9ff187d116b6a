#!/bin/bash
# Same-box A/B through bench.py itself (the driver's arguments and a long run), every variant interleaved.
# usage: bash tools/bench_ab.sh <rounds> spec...     spec = tag[:ENV=VAL,...]; tag "cur" = p25rx_amd/libp25fe.so, else
#        build/variants/libp25fe_<tag>.so
R=$1; shift
SPECS=("$@")
for r in $(seq 1 $R); do
  for S in "20 5" "400 20"; do
    read -r K W <<< "$S"
    for spec in "${SPECS[@]}"; do
      tag=${spec%%:*}; envs=""
      if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
      lib=$PWD/build/variants/libp25fe_$tag.so; [ "$tag" == "cur" ] && lib=$PWD/p25rx_amd/libp25fe.so
      line=$(env $envs P25FE_LIB=$lib python3 bench.py --no-extra --no-cpu --steps $K --warmup $W 2>/dev/null | tail -1)
      python3 - "$spec" "$K" "$line" <<'PY'
import json, sys
r = json.loads(sys.argv[3])
print("%-28s steps %4s: ms/step %.4f  K1 %.4f (n=%d)  serial %.4f  gate %s" % (sys.argv[1], sys.argv[2], r["ms_per_step"], r["roofline"]["kernel_ms"],
      r["roofline"]["kernel_ms_samples"], r["config"]["serial_ms_per_step"], r["config"]["parity_gate"][-5:]))
PY
    done
  done
done
