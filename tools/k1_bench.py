#!/usr/bin/env python3
"""Micro-benchmark of the kernels on random IQ (within-process A/B across env settings is done by the caller).
usage: k1_bench.py [seconds=600] [iters=20] [mode=k1|k0|chz|run|split] [channels=1] [fmt=cf32|u8] [symbol_clock=0|1]
mode split = run, plus the per-kernel split from the library's own HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd.frontend import FrontEnd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "k1"
C = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fmt = sys.argv[5] if len(sys.argv) > 5 else "cf32"
clock = int(sys.argv[6]) if len(sys.argv) > 6 else 0
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
if fmt == "cf32" and mode in ("run", "split"):      # the symbol receiver's cost depends on the signal: real C4FM
    from p25rx_amd import c4fm
    iq = torch.empty((C, n, 2), dtype=torch.float32, device=dev)
    for c in range(C):
        c4fm.synth_torch(n, seed=1 + c, device=dev, out=iq[c])
elif fmt == "cf32":
    iq = torch.randn((C, n, 2), dtype=torch.float32, device=dev) * 0.3
else:
    iq = torch.randint(0, 256, (C, n, 2), dtype=torch.uint8, device=dev)
fe = FrontEnd(n_channels=C, symbol_clock=clock)
bb = None
def step():
    global bb
    if mode == "k1":
        bb, nb = fe.demod_dev(iq, bb=bb)
    elif mode == "k0":                      # config 3 stage 0: the buffer is read as a 2.4 Msps capture
        bb, nb = fe.predecim_dev(iq, out=bb)
    elif mode == "chz":                     # channeliser: the buffer is one 2.4 Msps capture -> 192 channels
        bb, nb = fe.channelise_dev(iq[0], out=bb)
    else:
        fe.run_dev(iq)
for _ in range(5):
    step()
if mode == "split":
    fe.profile_enable(True)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
for e0, e1 in ev:          # back-to-back launches, no host sync in between (steady-state clocks)
    e0.record(); step(); e1.record()
torch.cuda.synchronize()
ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
med, mn, p90 = ts[len(ts) // 2], ts[0], ts[int(len(ts) * 0.9)]
bps = 8.8 if fmt == "cf32" else 2.8
if mode == "chz":
    bps = 8.0 + 192 * 8 / 10.0              # 8 B read + 192 channels x 8 B per 10 input samples
tag = " ".join("%s=%s" % (k[6:], os.path.basename(v)) for k, v in sorted(os.environ.items()) if k.startswith("P25FE_"))
print("%-6s C=%d n=%d %s [%s]: med %.4f min %.4f p90 %.4f ms -> %.1f Gsamples/s, %.0f GB/s"
      % (mode, C, n, fmt, tag, med, mn, p90, C * n / med / 1e6, C * n * bps / med / 1e6))
if mode == "split":
    ms, calls = fe.profile_read()
    print("       per-kernel ms (K1 K2 K3 K4): " + " ".join("%.4f" % (m / max(1, calls)) for m in ms))
