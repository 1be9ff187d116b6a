#!/usr/bin/env python3
"""Measurement: the step of bench.py replayed from a hipGraph (torch.cuda.CUDAGraph = stream capture) against the live
launches.  (a) one serial step (K1 -> K2 -> K3 -> K4) per graph; (b) G pipelined steps + join per graph (the two-stream
fork / join captured as graph edges; needs P25FE_EXT_EVENTS=0: events attached to dispatches are not capturable).
usage: python tools/graph_probe.py [seconds=600] [G=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd, parse_results

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
n = int(secs * 240000) // 8 * 8
iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
_, truth = c4fm.synth_torch(n, seed=1000, device=dev, snr_db=30.0, out=iq)
fe = FrontEnd()
dib, res = fe.run_dev(iq)
torch.cuda.synchronize()

def timed(f, k, finish=None):
    for _ in range(5): f()
    if finish: finish()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    if finish: finish()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

# settle the clock
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(16): fe.run_dev_pipelined(iq, dibits=dib, result=res)
    fe.join_dev(); torch.cuda.synchronize()
print("live pipelined  %.4f ms/step" % timed(lambda: fe.run_dev_pipelined(iq, dibits=dib, result=res), 400, fe.join_dev))
print("live serial     %.4f ms/step" % timed(lambda: fe.run_dev(iq, dibits=dib, result=res), 400))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fe.run_dev(iq, dibits=dib, result=res)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fe.run_dev(iq, dibits=dib, result=res)
    print("graph, 1 serial step per graph   %.4f ms/step" % timed(g.replay, 400))
    nd = int(parse_results(res)[0]["n_dibits"])
    k = min(nd, len(truth) - 24)
    print("   parity after replay:", bool((dib[0, :k].cpu().numpy() == truth[24:24 + k]).all()))
except Exception as e:
    print("graph (serial) failed:", type(e).__name__, str(e)[:200])
if os.environ.get("P25FE_EXT_EVENTS") == "0":
    try:
        fe2 = FrontEnd()
        d2, r2 = fe2.run_dev_pipelined(iq); fe2.join_dev(); torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            for _ in range(G):
                fe2.run_dev_pipelined(iq, dibits=d2, result=r2)
            fe2.join_dev()
        ms = timed(g2.replay, 400 // G)
        print("graph, %d pipelined steps + join per graph   %.4f ms/step" % (G, ms / G))
    except Exception as e:
        print("graph (pipelined) failed:", type(e).__name__, str(e)[:300])
