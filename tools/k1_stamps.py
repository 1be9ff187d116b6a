#!/usr/bin/env python3
"""Per-phase shader-clock time of K1's sub-tiles, measured inside the loaded kernel (measurement build -DP25FE_K1_STAMP,
selected with P25FE_LIB).  usage: P25FE_LIB=build/abl/libp25fe_stamp.so k1_stamps.py [seconds=600] [mode=run|lin|run_u8|lin_u8]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd import _lib, c4fm
from p25rx_amd.frontend import FrontEnd

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
modes = sys.argv[2:] or ["run"]
n = int(secs * 240000) // 8 * 8
dev = torch.device("cuda", 0)
iq = torch.empty((1, n, 2), dtype=torch.float32, device=dev)
c4fm.synth_torch(n, seed=1, device=dev, out=iq[0])
if os.environ.get("K1_ZERO"):              # data-dependent power: all-zero samples (DVFS check)
    iq.zero_()
iq8 = torch.clamp(torch.round((iq + 1.0) * 127.5), 0, 255).to(torch.uint8)
fe = FrontEnd(n_channels=1)
L = _lib.load()
rd = L.p25fe_debug_k1_stamps
rd.argtypes = [C.POINTER(C.c_uint64)]
names = ["window wait + staging", "output stores + next window request", "decimator", "d stores + channel filter",
         "d carry + discriminator", "boxcar + transpose stores", "transpose reads / sign words"]
bb = None
for m in modes:
    x = iq8 if m.endswith("u8") else iq
    f = (lambda: fe.run_dev(x)) if m.startswith("run") else (lambda: fe.demod_dev(x))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 10
    e0.record()
    for _ in range(R):
        f()
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    nsub = buf[7]
    tot = sum(buf[i] for i in range(7))
    print("mode %s: %.1f us per call, %d sub-tiles per call, %.0f clock ticks per sub-tile (wave view)" % (
        m, e0.elapsed_time(e1) * 1e3 / R, nsub // R, tot / nsub))
    for i in range(7):
        print("   %-40s %8.0f ticks  %5.1f %%" % (names[i], buf[i] / nsub, 100.0 * buf[i] / tot))
    nwg = buf[11]
    us = e0.elapsed_time(e1) * 1e3 / R
    life_us = buf[10] / nwg / 100.0
    nit = max(1, buf[12])
    print("   per work item: %.2f sub-tiles, prologue %.0f ticks, in sub-tiles %.0f; per workgroup: %.1f items, lifetime %.0f ticks = %.2f us (shader clock %.2f GHz)" % (
        nsub / nit, buf[8] / nit, tot / nit, nit / nwg, buf[9] / nwg, life_us, buf[9] / nwg / life_us / 1e3))
    print("   resident waves (sum of lifetimes / kernel time): %.0f of 2816 slots; %d workgroups per call" % (
        life_us * (nwg / R) / us, nwg // R))
