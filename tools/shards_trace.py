#!/usr/bin/env python3
"""The sharded step in ONE process (one-rank RCCL communicator through p25rx_amd/rccl.py), for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o tr -- python3 tools/shards_trace.py [gather=root] [pipelined]
then   python3 tools/shards_trace.py --summarise OUT   prints the timeline of one steady-state step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, glob, re
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    k1 = [i for i, r in enumerate(rows) if "k_frontend" in r[2] and (r[1] - r[0]) > 100000]
    a, b = k1[-3], k1[-2]                  # one whole step in steady state
    t0 = rows[a][0]
    prev_end = None
    for s, e, n in rows[a:b + 1]:
        m = re.search(r"(k_\w+|nccl\w+|rccl\w+|\w*[Kk]ernel\w*)", n)
        gap = "" if prev_end is None else "  (gap %6.1f)" % ((s - prev_end) / 1e3)
        print("  +%8.1f us  %8.1f us  %-50s%s" % ((s - t0) / 1e3, (e - s) / 1e3, (m.group(1) if m else n)[:50], gap))
        prev_end = e
    print("step period %.1f us" % ((rows[b][0] - t0) / 1e3))
    sys.exit(0)

import torch
from p25rx_amd import c4fm, rccl
from p25rx_amd._lib import RESULT_DTYPE
from p25rx_amd.frontend import FrontEnd
gather = sys.argv[1] if len(sys.argv) > 1 else "root"
n = 600 * 240000
fe = FrontEnd()
ss = rccl.ShardStep(fe, 0, 1, n, rccl.unique_id())
ss.prepare()
halo = fe.shard_halo()
buf = torch.zeros((halo + n, 2), dtype=torch.float32, device="cuda")
c4fm.synth_torch(n, seed=3, device=torch.device("cuda", 0), out=buf[halo:])
result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
dibits = torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device="cuda")
ss.comm_timing(int(os.environ.get("SHARDS_TRACE_TIMING", "0")))   # no event packets between the kernels of the traced steps
pipelined = len(sys.argv) > 2 and sys.argv[2] == "pipelined"   # p25fe_shard_step_pipelined: the chain behind K1 runs beside the next step's K1
own = torch.cuda.Stream() if os.environ.get("SHARDS_TRACE_OWN_STREAM") else None      # a user stream instead of the NULL stream
with torch.cuda.stream(own if own is not None else torch.cuda.current_stream()):
    for _ in range(300):
        ss.step(buf, dibits, result, gather=gather, pipelined=pipelined)
    ss.join()
torch.cuda.synchronize()
print("done", ss.comm_ms())
