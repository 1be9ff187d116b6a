#!/usr/bin/env python3
"""The two-stream step (p25fe_run_dev_pipelined) of configs[1] for a kernel trace, and the trace's timeline:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o tr -- python3 tools/pipe_timeline.py <symbol_clock 0|1|2> [serial]
    python3 tools/pipe_timeline.py --summarise OUT
prints, over the steady-state steps: K1's duration and the K1 -> K1 gap, every other kernel's duration (median / p90) and WHERE in the
K1 beside it it ran (start offset from that K1's start).  `serial` = p25fe_run_dev (one stream): the kernels' times ALONE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, glob, re
    import numpy as np
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda n: (re.search(r"(k_\w+)", n).group(1) if re.search(r"(k_\w+)", n) else n[:40]) + ("<gen>" if "ILb1" in n or "<true>" in n else "")
    k1 = [r for r in rows if "k_frontend" in r[2] and (r[1] - r[0]) > 100000]
    k1 = k1[len(k1) // 2:-2]                                         # steady state: the second half, without the last steps
    if len(k1) < 4:
        print("too few K1 launches"); sys.exit(1)
    lo, hi = k1[0][0], k1[-1][1]
    d = np.array([(e - s) / 1e3 for s, e, _ in k1])
    gaps = np.array([(k1[i + 1][0] - k1[i][1]) / 1e3 for i in range(len(k1) - 1)])
    per = np.array([(k1[i + 1][0] - k1[i][0]) / 1e3 for i in range(len(k1) - 1)])
    print("K1: %d launches  duration median %.1f p10 %.1f p90 %.1f us;  K1 end -> next K1 start median %.1f p90 %.1f us;  period median %.1f us"
          % (len(k1), np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(gaps), np.percentile(gaps, 90), np.median(per)))
    starts = np.array([s for s, _, _ in k1])
    ends = np.array([e for _, e, _ in k1])
    stat = {}
    for s, e, n in rows:
        if s < lo or e > hi or "k_frontend" in n:
            continue
        i = int(np.searchsorted(starts, s, side="right")) - 1       # the K1 that was running (or had last started) when this kernel began
        off = (s - starts[i]) / 1e3
        inside = s < ends[i]
        stat.setdefault(short(n), []).append(((e - s) / 1e3, off, inside))
    print("%-28s %6s %8s %8s %8s   %s" % ("kernel", "n", "median", "p90", "max", "start offset into the K1 beside it (median; share that began while a K1 ran)"))
    for n, v in sorted(stat.items(), key=lambda kv: np.median([x[1] for x in kv[1]])):
        du = np.array([x[0] for x in v]); of = np.array([x[1] for x in v]); ins = np.mean([x[2] for x in v])
        print("%-28s %6d %8.1f %8.1f %8.1f   +%.1f us  (%.0f %%)" % (n, len(v), np.median(du), np.percentile(du, 90), du.max(), np.median(of), 100 * ins))
    sys.exit(0)

import time, torch
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
dev = torch.device("cuda", 0)
n = 600 * 240000
iq, truth = c4fm.synth_torch(n, seed=1003, device=dev, snr_db=30.0, clock_ppm=150.0 if mode else 0.0)
fe = FrontEnd(symbol_clock=mode)
d, r = fe.run_dev(iq)
N = int(os.environ.get("PIPE_STEPS", "300"))
for rnd in range(2):
    t0 = time.perf_counter()
    for _ in range(N):
        if serial:
            fe.run_dev(iq, dibits=d, result=r)
        else:
            fe.run_dev_pipelined(iq, dibits=d, result=r)
    if not serial:
        fe.join_dev()
    torch.cuda.synchronize()
    print("mode %d %s: %.4f ms per step" % (mode, "serial" if serial else "pipelined", (time.perf_counter() - t0) / N * 1e3), flush=True)
