#!/usr/bin/env python3
"""Wall time per pipelined call (p25fe_run_dev_pipelined) of configs[1] generated with a 150 ppm sample clock, under a tracking symbol
clock.  usage: tracking_step.py [symbol_clock=2]   (P25FE_PIPE_DEPTH = scratch sets; profiles/r05_tracking_pipeline.txt)"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd, parse_results
dev = torch.device("cuda", 0)
n = 600 * 240000
iq, truth = c4fm.synth_torch(n, seed=1003, device=dev, snr_db=30.0, clock_ppm=150.0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fe = FrontEnd(symbol_clock=mode)
d, r = fe.run_dev(iq)
for _ in range(600):
    fe.run_dev_pipelined(iq, dibits=d, result=r)
fe.join_dev(); torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(300):
        fe.run_dev_pipelined(iq, dibits=d, result=r)
    fe.join_dev(); torch.cuda.synchronize()
    print("mode %d: %.4f ms per pipelined step" % (mode, (time.perf_counter() - t0) / 300 * 1e3), flush=True)
