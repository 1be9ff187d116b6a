#!/bin/bash
# The C-ABI shard step over a ONE-rank RCCL communicator (self send / recv) on one GPU: exact-byte gather (one host wait per step)
# against whole rows + compaction (no host wait).  usage: bash tools/shards_gather_ab.sh [seconds=600] [steps=200]
SECS=${1:-600}; STEPS=${2:-200}
python3 - "$SECS" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from p25rx_amd import c4fm
iq, _, _ = c4fm.synth(float(sys.argv[1]) if float(sys.argv[1]) <= 20 else 20.0, seed=5, snr_db=25.0)
reps = int(np.ceil(float(sys.argv[1]) / 20.0)) if float(sys.argv[1]) > 20 else 1
with open("/tmp/shards_cap.cf32", "wb") as f:
    for _ in range(reps):
        iq.tofile(f)
PY
for r in 1 2 3; do
  for g in exact rows; do
    ./build/p25fe_shards -n 1 -k $STEPS -g $g /tmp/shards_cap.cf32 /tmp/shards_dib.out | tail -1
  done
done
rm -f /tmp/shards_cap.cf32 /tmp/shards_dib.out
