"""GPU tests of the host-side mirrors of the reference interface (DemodTask / RecvTask shape) and of the
larger-than-one-LDS-chunk scan path."""
import os
import queue
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_demodtask_recvtask_pipeline(c4fm_1s):
    """Same wiring as src/main.rs:235-287: reader -> DemodTask -> RecvTask, u8 chunks of 32768 bytes."""
    from oracle import oracle as O
    from p25rx_amd import c4fm
    from p25rx_amd.consts import BUF_BYTES
    from p25rx_amd.demod import DemodTask
    from p25rx_amd.recv import RecvTask
    u8 = c4fm.to_u8(c4fm_1s[0])
    reader, hub, chan = queue.Queue(), queue.Queue(), queue.Queue()
    for o in range(0, len(u8), BUF_BYTES):
        reader.put(u8[o:o + BUF_BYTES])
    reader.put(None)
    DemodTask(reader, hub, chan).run()
    chan.put(None)
    got, syncs, dumped = [], [], []
    RecvTask(chan, lambda d, sp, sd: (got.append(d), syncs.append(sp))).run(cb=lambda s: dumped.append(s))
    od = O.Demod()
    bb = np.concatenate([od.feed_u8(u8[o:o + BUF_BYTES]) for o in range(0, len(u8), BUF_BYTES)])
    dib, spos, _ = O.Recv().feed(bb)
    assert np.array_equal(np.concatenate(got), dib)
    assert np.array_equal(np.concatenate(syncs), spos)
    assert np.array_equal(np.concatenate(dumped).view(np.uint32), bb.view(np.uint32))    # the -w dump hook (src/recv.rs:152)
    n_chunks = (len(u8) + BUF_BYTES - 1) // BUF_BYTES
    assert hub.qsize() == n_chunks // 4                                                  # Throttler::new(4), src/demod.rs:67


@pytest.mark.parametrize("mode", ["u8", "cf32", "bb"])
def test_cpp_replay_driver(tmp_path, c4fm_1s, mode):
    """C++ DemodTask/RecvTask mirror (p25rx_amd/host) through the replay-style CLI vs the oracle."""
    from oracle import oracle as O
    from p25rx_amd import c4fm
    exe = os.path.join(ROOT, "build", "p25fe_replay")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    iq = c4fm_1s[0]
    src = tmp_path / ("in." + mode)
    if mode == "u8":
        u8 = c4fm.to_u8(iq)
        u8.tofile(src)
        ref = O.Recv().feed(O.Demod().feed_u8(u8))[0]
    elif mode == "cf32":
        iq.tofile(src)
        ref = O.run_cf32(iq)
    else:
        bb = O.Demod().feed_cf32(iq)
        bb.tofile(src)
        ref = O.Recv().feed(bb)[0]
    out = tmp_path / "dibits.out"
    subprocess.check_call([exe, mode, str(src), str(out)])
    assert np.array_equal(np.fromfile(out, dtype=np.uint8), ref)


def test_scan_spans_several_lds_chunks():
    """K3 stages 16384 tile summaries in LDS at a time: 70 M baseband samples = 34180 tiles cross a chunk boundary."""
    import torch
    from oracle import oracle as O
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd, parse_results
    iq, _, _ = c4fm.synth(2.0, seed=9, snr_db=25.0, frame_dibits=1100)
    bb1 = O.Demod().feed_cf32(iq)[200:]                      # start mid-stream so the tiling repeats irregularly
    reps = 70_000_000 // len(bb1) + 1
    bb = np.tile(bb1, reps)[:70_000_000]
    r = O.Recv()
    dib, spos, sdib = r.feed(bb)
    t = torch.from_numpy(bb).cuda()
    fe = FrontEnd()
    d, res, sp, sd = fe.slice_dev(t, len(bb), sync_cap=len(spos) + 8)
    rr = parse_results(res)[0]
    assert int(rr["n_dibits"]) == len(dib) and int(rr["n_sync"]) == len(spos)
    assert np.array_equal(d[0, :len(dib)].cpu().numpy(), dib)
    assert np.array_equal(sp[0, :len(spos)].cpu().numpy(), spos)
    assert np.array_equal(sd[0, :len(spos)].cpu().numpy().astype(np.uint64), sdib)
    st = r.state()
    assert int(rr["anchor_out"]["s"]) == st["s"] and float(rr["anchor_out"]["hi"]) == np.float32(st["hi"])


def test_timeshard_step_device_world1(c4fm_1s):
    """The N > 1 bench path (TimeShard.step_device) degenerates to one shard at world = 1: same dibits as run_dev."""
    import torch
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd.sharding import TimeShard
    iq = c4fm_1s[0]
    n = len(iq) // 8 * 8
    fe = FrontEnd()
    ts = TimeShard(fe, 0, 1, n, None)
    ts.setup_device(torch, "cuda")
    buf = ts.alloc(torch, "cuda", torch.float32)
    buf[ts.halo:] = torch.from_numpy(iq[:n].view(np.float32).reshape(-1, 2)).cuda()
    result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    summ_all = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    dibits = torch.empty((1, n // 50 + 64), dtype=torch.uint8, device="cuda")
    off = ts.step_device(buf, result, summ_all, dibits)
    nd = int(parse_results(result)[0]["n_dibits"])
    ref, rres = FrontEnd().run_dev(buf[ts.halo:])
    assert nd == int(parse_results(rres)[0]["n_dibits"]) and int(off[0]) == 0
    assert torch.equal(dibits[0, :nd], ref[0, :nd])
