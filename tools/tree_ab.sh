#!/bin/bash
# Same-box, interleaved A/B of whole TREES through their own bench.py (the libraries' ABIs differ, so each tree runs its own
# driver): build/r5tree (git archive of round 5's last commit) and the working tree.
#   usage: bash tools/tree_ab.sh <rounds> [extras=0|1]     extras=1: the full default run (configs[3] / [4] on one GPU are extras)
R=${1:-2}; EX=${2:-0}
for r in $(seq 1 $R); do
  for t in build/r3tree build/r5tree .; do
    [ -f $t/bench.py ] || continue
    if [ "$EX" == "1" ]; then
      line=$(cd $t && python3 bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1)
    else
      line=$(cd $t && python3 bench.py --steps 400 --warmup 20 --no-cpu --no-extra 2>/dev/null | tail -1)
    fi
    python3 - "$t" "$line" <<'PY'
import json, sys
r = json.loads(sys.argv[2])
print("%-14s ms/step %.4f  K1 %.4f frac %.4f  serial %s  gate %s" % (sys.argv[1], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"],
      r["config"].get("serial_ms_per_step"), r["config"]["parity_gate"][-5:]))
for e in r.get("extra", []):
    n = e.get("config", "")
    if n.startswith("configs[3]") or n.startswith("configs[4]") or n.startswith("configs[2]") or n.startswith("configs[1] as u8 I/Q pairs (the"):
        print("      %-70s ms %.4f  kernel_ms %s  frac %s  gate %s" % (n[:70], e.get("ms_per_step", 0), e.get("kernel_ms"), e.get("frac"), e.get("parity_gate")))
PY
  done
done
