#!/usr/bin/env python3
"""K1 (planar, cf32, built-in numbers) alone, serial steps, the library's own HIP events -- a script that runs unchanged in
build/r3tree, build/r4tree and the working tree (it only uses FrontEnd.run_dev / run_dev_pipelined / profile_*).
usage (from a tree's root): python3 <path>/k1_tree_probe.py [seconds=600] [iters=300]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from p25rx_amd.frontend import FrontEnd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((n, 2), dtype=torch.float32, device=dev, generator=g) * 0.3
fe = FrontEnd()
dib, res = fe.run_dev(x)
for _ in range(600):
    fe.run_dev(x, dibits=dib, result=res)
torch.cuda.synchronize()
out = []
for mode in ("serial", "pipelined"):
    fe.profile_enable(3 if mode == "pipelined" else 1)
    for _ in range(iters):
        if mode == "serial":
            fe.run_dev(x, dibits=dib, result=res)
        else:
            fe.run_dev_pipelined(x, dibits=dib, result=res)
    fe.join_dev()
    torch.cuda.synchronize()
    ms, calls = fe.profile_read()
    fe.profile_enable(False)
    out.append("%s K1 %.4f ms (n=%d)" % (mode, ms[0] / max(calls, 1), calls))
print("%-16s %s" % (os.path.basename(os.getcwd()), "   ".join(out)), flush=True)
