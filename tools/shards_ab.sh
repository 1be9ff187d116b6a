#!/bin/bash
# Same-box A/B of the C-ABI shard step over a ONE-rank RCCL communicator (the root plays its own peer): the round-4 library
# (build/r4tree, `git archive f58eca9` built in place) against the working tree.  usage: bash tools/shards_ab.sh [seconds=600] [steps=300] [rounds=3]
SECS=${1:-600}; STEPS=${2:-300}; ROUNDS=${3:-3}
python3 - "$SECS" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from p25rx_amd import c4fm
secs = float(sys.argv[1])
iq, _, _ = c4fm.synth(min(secs, 20.0), seed=5, snr_db=25.0)
reps = int(np.ceil(secs / 20.0)) if secs > 20 else 1
with open("/tmp/shards_cap.cf32", "wb") as f:
    for _ in range(reps):
        iq.tofile(f)
PY
for r in $(seq 1 $ROUNDS); do
  if [ -x build/r4tree/build/p25fe_shards ]; then
    echo -n "r4  rows : "; ./build/r4tree/build/p25fe_shards -n 1 -k $STEPS -g rows /tmp/shards_cap.cf32 /tmp/shards_dib_old.out | tail -1
  fi
  echo -n "new rows : "; ./build/p25fe_shards -n 1 -k $STEPS -g rows /tmp/shards_cap.cf32 /tmp/shards_dib.out | tail -1
  echo -n "new rows, head by event wait : "; P25FE_SHARD_HEAD_WAIT=event ./build/p25fe_shards -n 1 -k $STEPS -g rows /tmp/shards_cap.cf32 /tmp/shards_dib3.out | tail -1
  echo -n "new rows, side stream at high priority : "; P25FE_SHARD_CS_PRIO=1 ./build/p25fe_shards -n 1 -k $STEPS -g rows /tmp/shards_cap.cf32 /tmp/shards_dib3.out | tail -1
  echo -n "new rows, events on every step : "; ./build/p25fe_shards -n 1 -k $STEPS -g rows -t 1 /tmp/shards_cap.cf32 /tmp/shards_dib.out | tail -1
  echo -n "new rows, pipelined (-p) : "; ./build/p25fe_shards -n 1 -k $STEPS -g rows -p /tmp/shards_cap.cf32 /tmp/shards_dib4.out | tail -1
  echo -n "new rows, pipelined, the step's own order on the receive stream (layout 1) : "; P25FE_SHARD_PIPE_LAYOUT=1 ./build/p25fe_shards -n 1 -k $STEPS -g rows -p /tmp/shards_cap.cf32 /tmp/shards_dib3.out | tail -1
  echo -n "new rows, a stream of the program's own (-s) : "; ./build/p25fe_shards -n 1 -k $STEPS -g rows -s /tmp/shards_cap.cf32 /tmp/shards_dib3.out | tail -1
  echo -n "new rows, pipelined, a stream of the program's own (-p -s) : "; ./build/p25fe_shards -n 1 -k $STEPS -g rows -p -s /tmp/shards_cap.cf32 /tmp/shards_dib3.out | tail -1
  echo -n "new exact: "; ./build/p25fe_shards -n 1 -k $STEPS -g exact /tmp/shards_cap.cf32 /tmp/shards_dib2.out | tail -1
  cmp /tmp/shards_dib.out /tmp/shards_dib2.out && cmp /tmp/shards_dib.out /tmp/shards_dib3.out && cmp /tmp/shards_dib.out /tmp/shards_dib4.out && { [ ! -f /tmp/shards_dib_old.out ] || cmp /tmp/shards_dib.out /tmp/shards_dib_old.out; } && echo "streams identical"
done
rm -f /tmp/shards_cap.cf32 /tmp/shards_dib.out /tmp/shards_dib2.out /tmp/shards_dib3.out /tmp/shards_dib4.out /tmp/shards_dib_old.out
