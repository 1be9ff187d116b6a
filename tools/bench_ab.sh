#!/bin/bash
# Same-box A/B of library builds through bench.py itself (the driver's arguments and a long run), interleaved.
# usage: bash tools/bench_ab.sh <rounds> tag...      (build/variants/libp25fe_<tag>.so)
R=$1; shift
TAGS=("$@")
for r in $(seq 1 $R); do
  for S in "20 5" "400 20"; do
    read -r K W <<< "$S"
    for tag in "${TAGS[@]}"; do
      line=$(P25FE_LIB=$PWD/build/variants/libp25fe_$tag.so python3 bench.py --no-extra --no-cpu --steps $K --warmup $W 2>/dev/null | tail -1)
      python3 - "$tag" "$K" "$line" <<'PY'
import json, sys
r = json.loads(sys.argv[3])
print("%-6s steps %4s: ms/step %.4f  K1 %.4f (n=%d)  serial %.4f" % (sys.argv[1], sys.argv[2], r["ms_per_step"], r["roofline"]["kernel_ms"],
      r["roofline"]["kernel_ms_samples"], r["config"]["serial_ms_per_step"]))
PY
    done
  done
done
