#!/usr/bin/env python3
"""Interleaved kernel A/B on one box: every round runs each requested mode once, so that rocprofv3's per-kernel
averages (tools/ab.sh) compare like with like (same box, same clocks, same thermal state).
usage: kbench.py [seconds=600] [rounds=30] [channels=1] mode...   modes: lin (K1, linear baseband), run (K1 planar ->
K2 -> K3 -> K4), lin_u8, run_u8, k0, chz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd.frontend import FrontEnd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
modes = sys.argv[4:] or ["lin", "run"]
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
from p25rx_amd import c4fm
iq = torch.empty((C, n, 2), dtype=torch.float32, device=dev)
for c in range(C):
    c4fm.synth_torch(n, seed=1 + c, device=dev, out=iq[c])
iq8 = None
if any(m.endswith("u8") for m in modes):
    iq8 = torch.clamp(torch.round((iq + 1.0) * 127.5), 0, 255).to(torch.uint8)
kw = {}
if os.environ.get("P25FE_KBENCH_TAPS"):                    # e.g. "31,41": other Kaiser designs of the two filters -> specialised kernels
    from scipy import signal as sps
    import numpy as np
    n1, n2 = (int(x) for x in os.environ["P25FE_KBENCH_TAPS"].split(","))
    kw = dict(decim_taps=sps.firwin(n1, 11000.0, window=("kaiser", 6.0), fs=240000.0).astype(np.float32).tolist(),
              chan_taps=sps.firwin(n2, 6500.0, window=("kaiser", 4.5), fs=48000.0).astype(np.float32).tolist(), specialize=1)
fe = FrontEnd(n_channels=C, **kw)
bb = None
def step(m):
    global bb
    if m == "lin":
        bb, _ = fe.demod_dev(iq, bb=bb)
    elif m == "lin_u8":
        bb, _ = fe.demod_dev(iq8, bb=bb)
    elif m == "run":
        fe.run_dev(iq)
    elif m == "run_u8":
        fe.run_dev(iq8)
    elif m == "k0":
        fe.predecim_dev(iq)
    elif m == "chz":
        fe.channelise_dev(iq[0])
# Pre-warm: the chip reaches its steady clock / power state only after ~100 ms of load (a trace of 40 rounds from idle sits
# entirely inside that ramp: K1<u8>, which is issue-bound, read 216 us there and 171 us after it; K1<cf32>, HBM-bound, the
# same in both -- profiles/r05_u8_trace_spread.txt).  The pre-warm launches are part of a rocprofv3 trace of this script;
# tools/prof_summary.py summarises the LAST `rounds` launches of every kernel (the line below tells it how many).
import time
prewarm_ms = float(os.environ.get("KBENCH_PREWARM_MS", "150"))
n_pre = 0
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < prewarm_ms or n_pre < 3:
    for m in modes:
        step(m)
    torch.cuda.synchronize()
    n_pre += 1
for _ in range(rounds):
    for m in modes:
        step(m)
torch.cuda.synchronize()
print("kbench done: %s x %d rounds, C=%d n=%d" % (modes, rounds, C, n))
print("kbench timed_rounds=%d prewarm_rounds=%d" % (rounds, n_pre))
