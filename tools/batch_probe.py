#!/usr/bin/env python3
"""Where does the 256-channel batch (configs[3]) lose against one long channel?  Times the resident call (K1 -> K2 -> K3 -> K4,
one stream) for shapes of equal and of growing total size, interleaved on one box, and prints picoseconds per input sample.
usage: batch_probe.py [rounds=5]      (shapes below; the capture is noise -- K1's time does not depend on the content)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd.frontend import FrontEnd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [(1, 600.0), (10, 60.0), (256, 2.34375), (1, 3600.0), (60, 60.0), (16, 225.0), (256, 14.0625), (256, 60.0), (64, 240.0), (4, 3840.0)]
dev = torch.device("cuda", 0)
big = torch.empty((256 * 60 * 240000, 2), dtype=torch.float32, device=dev)
big.normal_(0.0, 0.3)
fes, views = {}, {}
for C, secs in SHAPES:
    n = int(secs * 240000) // 8 * 8
    views[(C, secs)] = big[: C * n].view(C, n, 2)
    fes[(C, secs)] = FrontEnd(n_channels=C, device=0)
out = {}
def once(key):
    fe, iq = fes[key], views[key]
    st = out.get(key)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    d, r = fe.run_dev(iq, dibits=st[0] if st else None, result=st[1] if st else None)
    e1.record()
    torch.cuda.synchronize()
    out[key] = (d, r)
    return e0.elapsed_time(e1)
for key in SHAPES:                      # warm: scratch allocation, code objects
    once(key); once(key)
acc = {k: [] for k in SHAPES}
for _ in range(rounds):
    for key in SHAPES:
        acc[key].append(once(key))
print("%-22s %10s %12s %12s" % ("channels x seconds", "GB", "ms (median)", "ps / sample"))
for C, secs in SHAPES:
    n = int(secs * 240000) // 8 * 8
    v = sorted(acc[(C, secs)])
    med = v[len(v) // 2]
    print("%4d x %-15.5g %10.2f %12.4f %12.4f" % (C, secs, C * n * 8 / 1e9, med, med * 1e9 / (C * n)))
