#!/usr/bin/env python3
"""Follow-up to batch_probe.py: is the 29.5 GB shape slower because of WHERE its bytes are (address range) or because of HOW LONG
the chip streams (power state)?  Same 6.9 GB shape (60 channels x 60 s) at four offsets of one 29.5 GB allocation, then four of them
back to back without a synchronisation (the same 29.5 GB of traffic in one burst)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd.frontend import FrontEnd
dev = torch.device("cuda", 0)
C, n = 60, 60 * 240000
big = torch.empty((256 * n, 2), dtype=torch.float32, device=dev)
big.normal_(0.0, 0.3)
fe = FrontEnd(n_channels=C, device=0)
views = [big[q * 64 * n: q * 64 * n + C * n].view(C, n, 2) for q in range(4)]
st = None
def call(v):
    global st
    st = fe.run_dev(v, dibits=st[0] if st else None, result=st[1] if st else None)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for v in views: call(v)
torch.cuda.synchronize()
for rnd in range(4):
    one = [timed(lambda v=v: call(v)) for v in views]
    burst = timed(lambda: [call(v) for v in views])
    same = timed(lambda: [call(views[0]) for _ in range(4)])
    print("round %d: one call at offset 0/7.4/14.7/22.1 GB: %s ms; four offsets back to back: %.4f ms (sum of singles %.4f); offset 0 four times: %.4f ms"
          % (rnd, " ".join("%.4f" % x for x in one), burst, sum(one), same))
# the same four regions through a library read (torch.sum) and a library copy into one fixed destination: is it this kernel or the memory?
dst = torch.empty_like(views[0])
for rnd in range(2):
    rd = [timed(lambda v=v: v.sum()) for v in views]
    cp = [timed(lambda v=v: dst.copy_(v)) for v in views]
    print("library round %d: torch.sum %s ms; copy_ %s ms" % (rnd, " ".join("%.4f" % x for x in rd), " ".join("%.4f" % x for x in cp)))
