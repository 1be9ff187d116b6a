#!/bin/bash
# p25fe_replay's bulk mode (-W: reader thread + pinned blocks + p25fe_run_host_windows) on 600 s captures read from files,
# beside the chunked mode; the dibit files must be identical.
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from p25rx_amd import c4fm
iq, _, _ = c4fm.synth(20.0, seed=5, snr_db=25.0)
u8 = c4fm.to_u8(iq)
with open("/tmp/cap.cf32", "wb") as f, open("/tmp/cap.u8", "wb") as g:
    for _ in range(30):
        iq.tofile(f)
        u8.tofile(g)
PY
ls -la /tmp/cap.cf32 /tmp/cap.u8
for i in 1 2; do
  build/p25fe_replay -W 64M cf32 /tmp/cap.cf32 /tmp/d1.out
  build/p25fe_replay -W 64M u8 /tmp/cap.u8 /tmp/d2.out
done
echo "chunked mode (64 reads of 32 768 bytes per call):"
time build/p25fe_replay cf32 /tmp/cap.cf32 /tmp/d3.out
cmp /tmp/d1.out /tmp/d3.out && echo "bulk and chunked dibits identical"
rm -f /tmp/cap.cf32 /tmp/cap.u8 /tmp/d1.out /tmp/d2.out /tmp/d3.out
