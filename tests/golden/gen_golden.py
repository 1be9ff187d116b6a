#!/usr/bin/env python3
"""Write tests/golden/c4fm_seed7_u8.npz: a SELF-GENERATED golden vector.

The reference holds no fixtures for this path.  Input: 0.25 s of seeded C4FM as
RTL-SDR style u8 I/Q (120 KB).  Expected: the oracle's baseband (bit patterns), dibits and
sync positions at the time of generation.  Its job is to catch regressions of the oracle and
to give the GPU path a committed vector that does not depend on the generator's PRNG.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from p25rx_amd import c4fm              # noqa: E402

iq, truth, _ = c4fm.synth(0.25, seed=7, snr_db=25.0, frame_dibits=360)
u8 = c4fm.to_u8(iq)
d = O.Demod()
bb = np.concatenate([d.feed_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
dib, spos, sdib = O.Recv().feed(bb)
n = min(len(dib), len(truth) - 24)
assert np.array_equal(dib[:n], truth[24:24 + n]), "oracle must decode the modulator's symbols"
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c4fm_seed7_u8.npz"),
                    iq_u8=u8, bb_bits=bb.view(np.uint32), dibits=dib, sync_pos=spos, sync_dibit=sdib,
                    truth=truth)
print("wrote golden:", len(u8), "bytes in,", len(bb), "baseband,", n, "dibits,", len(spos), "syncs")
