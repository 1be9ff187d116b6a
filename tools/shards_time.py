#!/usr/bin/env python3
"""Wall time per sharded step in ONE process (one-rank RCCL communicator), plain and pipelined, on torch's default stream and on a
stream of its own.  usage: shards_time.py [steps=300] [rounds=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd import c4fm, rccl
from p25rx_amd._lib import RESULT_DTYPE
from p25rx_amd.frontend import FrontEnd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 600 * 240000
fe = FrontEnd()
ss = rccl.ShardStep(fe, 0, 1, n, rccl.unique_id())
ss.prepare()
halo = fe.shard_halo()
buf = torch.zeros((halo + n, 2), dtype=torch.float32, device="cuda")
c4fm.synth_torch(n, seed=3, device=torch.device("cuda", 0), out=buf[halo:])
result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
dibits = torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device="cuda")
ss.comm_timing(0)
own = torch.cuda.Stream()
def run(pipelined, stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(50):
            ss.step(buf, dibits, result, gather="root", pipelined=pipelined)
        ss.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ss.step(buf, dibits, result, gather="root", pipelined=pipelined)
        ss.join(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
for r in range(rounds):
    print("round %d: plain default %.4f | plain own-stream %.4f | pipelined default %.4f | pipelined own-stream %.4f  ms per step"
          % (r, run(False, None), run(False, own), run(True, None), run(True, own)), flush=True)
