#!/bin/sh
# tools/pin/run.sh <p25rx checkout> -- dump the reference's constants and golden vectors, then run the pin tests.
# Needs cargo (nightly, Readme.md:11-16) and network access for the reference's git dependencies; nothing in this
# repository's test or bench runs calls it.
set -eu
REF=${1:?usage: tools/pin/run.sh /path/to/p25rx}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT="$ROOT/tests/golden/pin"
mkdir -p "$REF/src/bin" "$OUT"
cp "$HERE/dump_consts.rs" "$HERE/dump_golden.rs" "$REF/src/bin/"
# the capture: written by this repository (no radio needed)
( cd "$ROOT" && python3 -c "
import sys; sys.path.insert(0, 'tools/pin')
import make_standin
u8, dibits = make_standin.pin_capture()
u8.tofile('$OUT/pin_seed7.u8'); dibits.tofile('$OUT/pin_seed7.dibits')" )
( cd "$REF" && cargo run --release --bin dump_consts > "$OUT/consts.json" )
( cd "$REF" && cargo run --release --bin dump_golden -- "$OUT/pin_seed7.u8" "$OUT/baseband.f32le" "$OUT/nid.jsonl" )
python3 "$HERE/load_pin.py" "$OUT/consts.json"
cd "$ROOT" && python3 -m pytest tests/test_pin.py -q -m pin -rs
