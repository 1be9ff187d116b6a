#!/usr/bin/env python3
"""configs[3]'s placement penalty (profiles/r05_batch_probe.txt): the SAME K1 launch on the same 6.9 GB shape (60 channels x 60 s) costs 1.52 ms
in the lower half of a 29.5 GB allocation and 1.63 - 1.69 ms in the upper.  This script runs that launch on ONE region, for a counter pass:
    rocprofv3 --pmc <counters> -d OUT -o pmc --output-format csv -- python3 tools/batch_pmc.py <region 0..3> [separate]
`separate`: the four regions are four allocations of their own instead of views of one (remedy 1).  Prints the launch's time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd.frontend import FrontEnd
dev = torch.device("cuda", 0)
q = int(sys.argv[1]) if len(sys.argv) > 1 else 0
separate = len(sys.argv) > 2 and sys.argv[2] == "separate"
C, n = 60, 60 * 240000
if separate:
    bufs = [torch.empty((64 * n, 2), dtype=torch.float32, device=dev) for _ in range(4)]
    for b in bufs:
        b.normal_(0.0, 0.3)
    views = [b[:C * n].view(C, n, 2) for b in bufs]
else:
    big = torch.empty((256 * n, 2), dtype=torch.float32, device=dev)
    big.normal_(0.0, 0.3)
    views = [big[k * 64 * n: k * 64 * n + C * n].view(C, n, 2) for k in range(4)]
fe = FrontEnd(n_channels=C, device=0)
st = fe.demod_planar_only(views[q]) if hasattr(fe, "demod_planar_only") else None
d = r = None
for _ in range(3):
    d, r = fe.run_dev(views[q], dibits=d, result=r)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); d, r = fe.run_dev(views[q], dibits=d, result=r); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("region %d (%s): K1..K4 %s ms, ptr 0x%x" % (q, "separate allocations" if separate else "one 29.5 GB allocation",
                                                 " ".join("%.4f" % t for t in ts), views[q].data_ptr()))
