#!/usr/bin/env python3
"""Write tests/golden/dense_planes.npz: a SELF-GENERATED golden vector for the receiver (SPEC 3.7 / 3.8), baseband in, dibits out.

The ten polyphase planes of the baseband are independent sample sets: every plane carries its own back-to-back frame-sync words,
plane r's ending at positions 240 m + 21 r -- ten detections per 240 samples, 320 per 7 680-sample tile, denser than any transmitter
and close to SPEC 3.7's bound.  Input: 2 tiles + 333 samples of that baseband with a little seeded noise (float32 bit patterns,
63 KB).  Expected: the oracle's dibits, sync positions and sync dibit indices at the time of generation, for the three symbol
clocks, without and with lock drops.  Pins the oracle across rounds on the scene real traffic never produces; the GPU tests compare the
HIP path with the same file (no PRNG between the vector and its consumers)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402

DROPS = [1000, 1007, 5000, 7685, 12001]
pat = np.array([1.0 if (0x050cdf >> j) & 1 else -1.0 for j in range(24)], dtype=np.float32)
n_bb = 2 * 7680 + 333
q = np.arange(n_bb, dtype=np.int64)
bb = (0.24 * pat[((q - 21 * (q % 10) + 230) // 10) % 24]).astype(np.float32)
bb += (0.004 * np.random.default_rng(11).standard_normal(n_bb)).astype(np.float32)
out = {"bb_bits": bb.view(np.uint32), "drops": np.array(DROPS, dtype=np.int64)}
for mode in (0, 1, 2):
    for tag, drops in (("", []), ("_drops", DROPS)):
        d, sp, sd = O.recv_range(bb, O.make_config(symbol_clock=mode), drops)
        assert len(sp) > 600
        out["dibits_m%d%s" % (mode, tag)] = d
        out["sync_pos_m%d%s" % (mode, tag)] = sp
        out["sync_dibit_m%d%s" % (mode, tag)] = sd.astype(np.uint64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dense_planes.npz"), **out)
print("wrote golden:", n_bb, "baseband samples,", {k: len(v) for k, v in out.items() if k.startswith("sync_pos")})
