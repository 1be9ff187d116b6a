#!/bin/bash
# A/B of the two-stream step on one box: every variant interleaved, the driver's exact bench arguments and a long run.
# usage (GPU box): bash tools/pipe_ab.sh <out-dir> [rounds]
OUT=${1:-gpurun_out/pipe_ab}; ROUNDS=${2:-3}
mkdir -p "$OUT"
run() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --no-extra --no-cpu "$@" 2>>"$OUT/err.log" | tail -1 >> "$OUT/$name.jsonl"
}
for r in $(seq 1 "$ROUNDS"); do
  for S in "20 5" "400 20"; do
    set -- $S
    run "serial_$1"        X=1 -- --steps $1 --warmup $2 --no-pipeline
    run "rec_$1"           P25FE_EXT_EVENTS=0 -- --steps $1 --warmup $2
    run "ext_$1"           P25FE_EXT_EVENTS=1 -- --steps $1 --warmup $2
    run "ext_cu8_$1"       P25FE_EXT_EVENTS=1 P25FE_RX_CUS=8 -- --steps $1 --warmup $2
    run "ext_cu16_$1"      P25FE_EXT_EVENTS=1 P25FE_RX_CUS=16 -- --steps $1 --warmup $2
    run "ext_cu32_$1"      P25FE_EXT_EVENTS=1 P25FE_RX_CUS=32 -- --steps $1 --warmup $2
    run "ext_cu64_$1"      P25FE_EXT_EVENTS=1 P25FE_RX_CUS=64 -- --steps $1 --warmup $2
  done
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f) if l.strip().startswith("{")]
    print("%-14s ms/step %s | K1 %s | serial %s" % (os.path.basename(f)[:-6], [r["ms_per_step"] for r in rows],
          [r["roofline"]["kernel_ms"] for r in rows], [r["config"].get("serial_ms_per_step") for r in rows]))
PY
