"""debug (test infrastructure: imports the oracle): run_dev / slice paths against the oracle receiver, channel by channel (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd, parse_results

c = int(sys.argv[1]) if len(sys.argv) > 1 else 37
iq, _, _ = c4fm.synth(0.2, seed=500 + c, snr_db=12.0 + (c % 20), freq_offset_hz=20.0 * (c % 11) - 100.0,
                      timing_offset=c % 50, frame_dibits=300 + 7 * (c % 13))
bb = O.Demod().feed_cf32(iq)
rx = O.Recv()
out = rx.feed(bb)
print("oracle:", type(out), [type(x) for x in out] if isinstance(out, tuple) else "")
ref_dib, ref_pos, ref_sd = out[0], out[1], out[2]
print("oracle n_dibits", len(ref_dib), "sync_pos", list(ref_pos), "sync_dibit", list(ref_sd))
fe = FrontEnd(device=0)
t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
dib, res = fe.run_dev(t)
r = parse_results(res)[0]
print("run_dev n_dibits", int(r["n_dibits"]), "n_sync", int(r["n_sync"]), "anchor", r["anchor_out"], "first_event", int(r["first_event"]))
got = dib[0, :int(r["n_dibits"])].cpu().numpy()
k = min(len(got), len(ref_dib))
d = np.nonzero(got[:k] != ref_dib[:k])[0]
print("first diffs", d[:10])
# slice path on the oracle's baseband (linear -> planarize)
fe2 = FrontEnd(device=0)
o2 = fe2.slice(bb, sync_cap=64)
pass
print(" n_dibits", len(o2[0]), "sync_pos", list(o2[1]), "sync_dibit", list(o2[2]))
# baseband parity of the planar K1 path is implicit; check the linear one
bbg, nb = fe.demod_dev(t)
print("linear baseband equal:", np.array_equal(bbg[0, :nb].cpu().numpy().view(np.uint32), bb.view(np.uint32)))
