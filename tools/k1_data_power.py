#!/usr/bin/env python3
"""Is K1's time set by its instruction stream or by the chip's power limit?  The SAME kernel (same instructions, same
addresses, same launch) on inputs that differ only in how many bits toggle: constant bytes, a real C4FM capture, uniform
random bytes / floats.  usage: k1_data_power.py [seconds=600] [iters=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
sig, _ = c4fm.synth_torch(n, seed=1, device=dev)
cases = [("u8 constant 127", torch.full((n, 2), 127, dtype=torch.uint8, device=dev)),
         ("u8 C4FM capture", torch.clamp(torch.round((sig + 1.0) * 127.5), 0, 255).to(torch.uint8)),
         ("u8 uniform random", torch.randint(0, 256, (n, 2), dtype=torch.uint8, device=dev)),
         ("cf32 zeros", torch.zeros((n, 2), dtype=torch.float32, device=dev)),
         ("cf32 C4FM capture", sig),
         ("cf32 normal random", torch.randn((n, 2), dtype=torch.float32, device=dev) * 0.3)]
fe = FrontEnd()
for rnd in range(2):
    for name, x in cases:
        for _ in range(10):
            fe.run_dev(x)
        fe.profile_enable(1)
        for _ in range(iters):
            fe.run_dev(x)
        torch.cuda.synchronize()
        ms, calls = fe.profile_read()
        fe.profile_enable(False)
        print("%-20s K1 %.4f ms   (K2 %.4f K3 %.4f K4 %.4f)" % (name, ms[0] / calls, ms[1] / calls, ms[2] / calls, ms[3] / calls), flush=True)
