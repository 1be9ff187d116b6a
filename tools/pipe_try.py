#!/usr/bin/env python3
"""Feasibility experiment: one capture cut into S time shards that run on two HIP streams of ONE GPU, so that
shard i+1's K1 (HBM bound) overlaps shard i's K2/K3 (VALU / latency bound).  usage: pipe_try.py [seconds] [S] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd, parse_results, n_baseband
from p25rx_amd._lib import RESULT_DTYPE

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
n = int(secs * 240000) // (8 * S) * (8 * S)
iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
_, truth = c4fm.synth_torch(n, seed=1000, device=dev, out=iq)
ns = n // S
fes = [FrontEnd() for _ in range(S)]
halo = fes[0].shard_halo()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
res = [torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(S)]
summ_all = torch.empty((S, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
bb0 = [n_baseband(0, k * ns) for k in range(S)]
bbn = [n_baseband(k * ns, ns) for k in range(S)]
d_bb0 = torch.tensor(bb0, dtype=torch.int64, device=dev)
d_bbn = torch.tensor(bbn, dtype=torch.int64, device=dev)
cap = (ns // 50 + 64 + 15) // 16 * 16
dibs = [torch.empty((1, cap), dtype=torch.uint8, device=dev) for _ in range(S)]
anchors = offsets = None
ev_done = [torch.cuda.Event() for _ in range(S)]
main = torch.cuda.current_stream()

def step():
    global anchors, offsets
    ev0 = torch.cuda.Event(); ev0.record(main)
    for k in range(S):
        st = streams[k % 2]
        st.wait_event(ev0)
        with torch.cuda.stream(st):
            h = halo if k else 0
            fes[k].shard_pass1(iq[k * ns - h:(k + 1) * ns], offset=h, n_hist=h, abs0=k * ns, result=res[k])
            summ_all[k].copy_(res[k][0])
            ev_done[k].record(st)
    for k in range(S):
        main.wait_event(ev_done[k])
    anchors, offsets = fes[0].shard_resolve_dev(summ_all, d_bb0, d_bbn, anchors, offsets)
    ev1 = torch.cuda.Event(); ev1.record(main)
    for k in range(S):
        st = streams[k % 2]
        st.wait_event(ev1)
        with torch.cuda.stream(st):
            fes[k].shard_pass2(anchors[k:k + 1], bbn[k], dev, result=res[k], dibits=dibs[k])
            ev_done[k].record(st)
    for k in range(S):
        main.wait_event(ev_done[k])

fe1 = FrontEnd()
def step_ref():
    fe1.run_dev(iq)

for fn, name in ((step_ref, "single pass"), (step, "S=%d shards on 2 streams" % S)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%-28s %.4f ms/step  %.1f Gsamples/s" % (name, dt * 1e3, n / dt / 1e9))
# parity: concatenated shard dibits == single pass
dref, rref = fe1.run_dev(iq)
ref = dref[0, :int(parse_results(rref)[0]["n_dibits"])].cpu().numpy()
got = np.concatenate([dibs[k][0, :int(parse_results(res[k])[0]["n_dibits"])].cpu().numpy() for k in range(S)])
print("parity with single pass:", np.array_equal(ref, got), len(ref), len(got))
