#!/usr/bin/env python3
"""K1 (planar, built-in numbers) time against resident waves per CU: the process's P25FE_K1_LDS_PAD pads every K1 workgroup's
LDS so that only k one-wave workgroups fit a CU; P25FE_LIB selects the product or a truncated measurement build (load
pipeline only / arithmetic on cache-resident loads).  One line per format.  docs/K1_MODEL.md uses the table.
usage: P25FE_K1_LDS_PAD=<bytes> [P25FE_LIB=...] k1_occupancy.py <label> [seconds=600] [iters=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd

label = sys.argv[1] if len(sys.argv) > 1 else "cur"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
sig, _ = c4fm.synth_torch(n, seed=1, device=dev)
u8 = torch.clamp(torch.round((sig + 1.0) * 127.5), 0, 255).to(torch.uint8)
fe = FrontEnd()
pad = int(os.environ.get("P25FE_K1_LDS_PAD", "0"))
for name, x in (("cf32", sig), ("u8", u8)):
    dib, res = fe.run_dev(x)
    for _ in range(60):
        fe.run_dev(x, dibits=dib, result=res)
    fe.profile_enable(1)
    for _ in range(iters):
        fe.run_dev(x, dibits=dib, result=res)
    torch.cuda.synchronize()
    ms, calls = fe.profile_read()
    fe.profile_enable(False)
    print("%-6s pad %6d  waves/CU %2d  %-5s K1 %.4f ms" % (label, pad, 163840 // (13076 + pad), name, ms[0] / calls), flush=True)
