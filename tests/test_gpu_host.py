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
    n_events = chan.qsize() - 1
    RecvTask(chan, lambda d, sp, sd: (got.append(d), syncs.append(sp)), hub=hub).run(cb=lambda s: dumped.append(s))
    od = O.Demod()
    bb = np.concatenate([od.feed_u8(u8[o:o + BUF_BYTES]) for o in range(0, len(u8), BUF_BYTES)])
    dib, spos, _ = O.Recv().feed(bb)
    assert np.array_equal(np.concatenate(got), dib)
    assert np.array_equal(np.concatenate(syncs), spos)
    assert np.array_equal(np.concatenate(dumped).view(np.uint32), bb.view(np.uint32))    # the -w dump hook (src/recv.rs:152)
    n_chunks = (len(u8) + BUF_BYTES - 1) // BUF_BYTES
    evs = [hub.get() for _ in range(hub.qsize())]
    assert sum(e.kind == "UpdateSignalPower" for e in evs) == n_chunks // 4              # Throttler::new(4), src/demod.rs:67
    st = [e.value for e in evs if e.kind == "UpdateStats"]
    assert len(st) == n_events // 16                                                     # Throttler::new(16), src/recv.rs:141
    assert all(x["dibits"] <= len(dib) and x["syncs"] <= len(spos) for x in st)           # (15 chunks in this second: none yet)


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
    for batch in ("1", "64"):                                 # any chunking of the file gives the same stream
        subprocess.check_call([exe, "-b", batch, mode, str(src), str(out)])
        assert np.array_equal(np.fromfile(out, dtype=np.uint8), ref), batch


def test_cpp_record_replay_roundtrip_and_events(tmp_path):
    """The reference's -w / -r pair (src/main.rs:95-102, 162-169, 278-283): record the baseband while receiving,
    replay the recording, get the same dibits; plus the hub-vocabulary JSON lines (sigPower, nid, updateStats)."""
    import json
    from oracle import oracle as O
    from p25rx_amd import c4fm
    from p25rx_amd.consts import BUF_BYTES
    exe = os.path.join(ROOT, "build", "p25fe_replay")
    iq, _, info = c4fm.synth(1.5, seed=21, snr_db=18.0, nid=lambda f: (0x293, 0x5))
    u8 = c4fm.to_u8(iq)
    src, bbf, d1, d2, js = (tmp_path / n for n in ("in.u8", "bb.f32le", "d1.out", "d2.out", "ev.jsonl"))
    u8.tofile(src)
    err = subprocess.run([exe, "-b", "1", "-w", str(bbf), "-j", str(js), "u8", str(src), str(d1)], check=True,
                         capture_output=True, text=True).stderr
    # the recording is the oracle's baseband, bit for bit, and replaying it reproduces the dibits
    od = O.Demod()
    bb = np.concatenate([od.feed_u8(u8[o:o + BUF_BYTES]) for o in range(0, len(u8), BUF_BYTES)])
    rec = np.fromfile(bbf, dtype=np.float32)
    assert np.array_equal(rec.view(np.uint32), bb.view(np.uint32))
    subprocess.check_call([exe, "bb", str(bbf), str(d2)])
    dib, spos, sdib = O.Recv().feed(bb)
    assert np.array_equal(np.fromfile(d1, dtype=np.uint8), dib)
    assert np.array_equal(np.fromfile(d2, dtype=np.uint8), dib)
    # events: one power report per 4 chunks (Throttler::new(4), src/demod.rs:67), one nid per frame sync, one stats row
    ev = [json.loads(l) for l in open(js)]
    n_chunks = (len(u8) + BUF_BYTES - 1) // BUF_BYTES
    assert sum(e["event"] == "sigPower" for e in ev) == n_chunks // 4
    assert "%d power reports" % (n_chunks // 4) in err
    nids = [e for e in ev if e["event"] == "nid"]
    ref = O.nid_decode(dib, sdib, spos)
    assert [n["sync_pos"] for n in nids] == [int(x) for x in spos]
    assert [(n["nac"], n["duid"], n["errors"], n["valid"]) for n in nids] == \
           [(int(r["nac"]), int(r["duid"]), int(r["n_errors"]), int(r["valid"])) for r in ref]
    assert any(n["valid"] == 1 and n["nac"] == 0x293 and n["duid"] == 0x5 for n in nids)
    per = [e for e in ev if e["event"] == "updateStats" and e.get("periodic")]
    assert len(per) == n_chunks // 16                             # Throttler::new(16), src/recv.rs:141, 162-165
    assert all(a["dibits"] <= b["dibits"] for a, b in zip(per, per[1:])) and per[-1]["dibits"] <= len(dib)
    st = [e for e in ev if e["event"] == "updateStats" and not e.get("periodic")]
    assert len(st) == 1 and st[0]["dibits"] == len(dib) and st[0]["syncs"] == len(spos)
    assert st[0]["bch"]["totalWords"] == sum(int(r["valid"]) >= 0 for r in ref)


def test_scan_spans_several_lds_chunks():
    """K3 stages 4096 tile summaries in LDS at a time and carries the receiver state from chunk to chunk: 70 M baseband
    samples = 9115 tiles of 7680 = three chunks (and the linear -> polyphase conversion of a long stream)."""
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


def _shared_gpu_worker(rank, world, port, n_per, q):
    import torch
    import torch.distributed as dist
    from p25rx_amd import c4fm
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd.sharding import HostStagedComm, TimeShard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        iq, _, _ = c4fm.synth(world * n_per / 240000.0, seed=55, snr_db=22.0, frame_dibits=900)
        fe = FrontEnd()
        ts = TimeShard(fe, rank, world, n_per, dist, comm=HostStagedComm(dist, rank, world))
        ts.setup_device(torch, "cuda")
        buf = ts.alloc(torch, "cuda", torch.float32)
        buf[ts.halo:] = torch.from_numpy(iq[rank * n_per:(rank + 1) * n_per].view(np.float32).reshape(-1, 2)).cuda()
        result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        summ_all = torch.empty((world, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        dibits = torch.zeros((1, ts.dibit_cap), dtype=torch.uint8, device="cuda")
        for _ in range(2):                                      # twice: scratch and profiling slots are reused
            off = ts.step_device(buf, result, summ_all, dibits)             # dibits gathered to rank 0 (the default)
        total = int(off[-1])
        root_stream = ts.d_stream[:total].cpu().numpy() if rank == 0 else None
        ts.d_stream.zero_()
        off = ts.step_device(buf, result, summ_all, dibits, gather="all")   # all-gather form: the stream on every rank
        q.put((rank, off.cpu().numpy().tolist(), ts.d_stream[:total].cpu().numpy(),
               int(parse_results(result)[0]["n_dibits"]), root_stream))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_timeshard_step_device_two_ranks_one_gpu():
    """The bench's N > 1 step with the REAL kernels: two processes share the GPU, the collectives are staged through
    the host (gloo) by HostStagedComm; everything else -- pass 1 split around the halo exchange, device resolve, pass 2,
    dibit all-gather and compaction -- is the product path.  The stream on every rank equals the oracle's."""
    import socket
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from p25rx_amd import c4fm
    world, n_per = 2, 240000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, n_per, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    iq, _, _ = c4fm.synth(world * n_per / 240000.0, seed=55, snr_db=22.0, frame_dibits=900)
    ref = O.run_cf32(iq)
    for rank, off, stream, nd, root_stream in got:
        assert off[-1] == len(ref) and off[rank + 1] - off[rank] == nd
        assert np.array_equal(stream, ref)
        assert rank != 0 or np.array_equal(root_stream, ref)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_time_shards_host_staged(scaling):
    """bench.py's N > 1 step (one capture cut into time shards, halo behind K1, device resolve, dibit gather) launched exactly
    as the driver launches it, with the collectives staged through gloo (P25FE_BENCH_HOST_STAGED: two ranks cannot share
    one GPU under RCCL).  Both gates must hold and the line must say what it measured."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, P25FE_BENCH_HOST_STAGED="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--seconds", "60", "--strong-seconds", "40"]
                         + (["--scaling", "strong", "--gather", "root_exact"] if scaling == "strong" else []),     # weak + rows are the defaults (the driver's command)
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 3
    assert d["config"]["parity_gate"].endswith("True") and d["config"]["gather_gate"].endswith("True")
    # the line names the gather mode the library executed (the shared-memory hook runs the exact-bytes form too)
    assert ("exactly the valid bytes" if scaling == "strong" else "whole rows") in d["config"]["sharding"]
    # weak: 60 s per rank = one 120 s capture; strong: the 60 s are the whole capture
    assert ("ONE 120 s capture" if scaling == "weak" else "ONE 60 s capture") in d["config"]["workload"] and d["value"] > 0
    # the library's own evidence of the job (p25fe_shard_info from every rank): here the shared-memory hook, i.e. no communicator
    ev = d["config"]["rccl"]
    assert ev["staged_test_hook"] and ev["rccl_ranks"] == [0, 0] and ev["communicators"] == [0] and len(ev["pci_bus_id_of_rank"]) == 2
    assert ev["distinct_gpus"] == 1 and not ev["broken"]
    if scaling == "weak":
        # configs[4] as worded rides along as an extra row of the default weak line, with its own gates
        (row,) = d["extra"]
        assert row["scaling"] == "strong" and "ONE 40 s capture" in row["config"] and row["parity_gate"].endswith("True")
        assert row["gather_gate"].endswith("True") and row["value"] > 0 and row["rccl"]["staged_test_hook"]
    else:
        assert "extra" not in d


@pytest.mark.timeout(300)
def test_bench_n2_exits_nonzero_on_a_false_gate():
    """A wrong multi-GPU result must not land in the driver's SCALE record as rc 0: with a fault injected into the last rank's
    dibits after the timed steps (TEST HOOK) the gather gate reads False in the JSON line and every rank exits non-zero."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(P25FE_BENCH_HOST_STAGED="1", P25FE_BENCH_INJECT_FAULT="gather")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--seconds", "20"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode != 0
    line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1 and json.loads(line[0])["config"]["gather_gate"].endswith("False")


@pytest.mark.parametrize("mode", [0, 1])
def test_run_host_windows_equals_one_pass(mode, tmp_path):
    """p25fe_run_host_windows (a long HOST capture as a pipeline of windows: H2D copy | kernels | dibits back): filter and
    receiver state cross the window boundaries like a shard's (halo + anchor), so any window size gives the dibits of one
    resident pass -- cf32 and u8, pageable (staged) and pinned (copied from directly) input, both symbol clocks, lock kept
    and handed on to the streaming calls, and the C++ replay driver's bulk mode on a file."""
    import torch
    from oracle import oracle as O
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd
    iq, _, _ = c4fm.synth(6.0, seed=61, snr_db=20.0, frame_dibits=600, clock_ppm=120.0 if mode else 0.0)
    cfg = O.make_config(symbol_clock=mode)
    ref = O.Recv(cfg).feed(O.Demod().feed_cf32(iq))[0]
    u8 = c4fm.to_u8(iq)
    ref8 = O.Recv(cfg).feed(O.Demod().feed_u8(u8))[0]
    for window in (8192, 100000, 1 << 19, 0):                        # many windows ... one (0 = the library's 64 MB)
        fe = FrontEnd(symbol_clock=mode)
        got, st = fe.run_host_windows(iq, window=window)
        assert np.array_equal(got, ref), window
        if window:
            assert st["n_windows"] == -(-len(iq) // (window // 8 * 8)) and not st["pinned_input"] and st["ms_h2d"] > 0
    fe = FrontEnd(symbol_clock=mode)
    got8, _ = fe.run_host_windows(u8, window=65536)
    assert np.array_equal(got8, ref8)
    # pinned input: the copy engine reads the caller's memory
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).pin_memory()
    fe = FrontEnd(symbol_clock=mode)
    got, st = fe.run_host_windows(t, window=1 << 18)
    assert st["pinned_input"] and np.array_equal(got, ref)
    # the call continues a stream and leaves it ready for more: chunked call | windows | chunked call == one pass
    fe = FrontEnd(symbol_clock=mode)
    a, b = 123457, 1000003
    parts = [fe.run_cf32(iq[:a]), fe.run_host_windows(iq[a:b], window=200000)[0], fe.run_cf32(iq[b:])]
    assert np.array_equal(np.concatenate(parts), ref)
    # ... and the baseband tail it leaves serves a later p25fe_slice
    fe = FrontEnd(symbol_clock=mode)
    d1, _ = fe.run_host_windows(iq[:b], window=300000)
    bb_rest = O.Demod().feed_cf32(iq)[len(O.Demod().feed_cf32(iq[:b])):]
    d2 = fe.slice(bb_rest)[0]
    assert np.array_equal(np.concatenate([d1, d2]), ref)
    # a remainder too short for a baseband sample of its own (n = 2 windows + 3): it rides with the window in front of it, and the
    # tail the call leaves is still the capture's last 256 baseband samples (it used to keep the tail from BEFORE the call)
    w = 300000 // 8 * 8
    for extra in (3, 7, 8, 9):
        m = 2 * w + extra
        fe = FrontEnd(symbol_clock=mode)
        d1, st = fe.run_host_windows(iq[:m], window=w)
        assert st["n_windows"] == (2 if extra < 8 else 3), (extra, st)
        bb_rest = O.Demod().feed_cf32(iq)[len(O.Demod().feed_cf32(iq[:m])):]
        d2 = fe.slice(bb_rest)[0]
        assert np.array_equal(np.concatenate([d1, d2]), ref), extra
    # lock drops INSIDE the capture (p25fe_resync_at_dev: absolute baseband indices): every window sees the list
    bb_all = O.Demod().feed_cf32(iq)
    drops = np.array([len(bb_all) // 5, len(bb_all) // 2 + 3, len(bb_all) - 4000], dtype=np.int64)
    r = O.Recv(cfg)
    pieces, prev = [], 0
    for q in list(drops) + [len(bb_all)]:
        pieces.append(r.feed(bb_all[prev:q])[0])
        r.resync()
        prev = q
    fe = FrontEnd(symbol_clock=mode)
    fe.resync_at_dev(torch.from_numpy(drops).cuda())
    got, _ = fe.run_host_windows(iq, window=150000)
    assert np.array_equal(got, np.concatenate(pieces))
    if mode == 0:
        # two channels, channel-major
        iq2, _, _ = c4fm.synth(6.0, seed=62, snr_db=20.0, frame_dibits=600)
        n = min(len(iq), len(iq2))
        both = np.stack([iq[:n], iq2[:n]])
        fe = FrontEnd(n_channels=2)
        outs, _ = fe.run_host_windows(both, window=250000)
        assert np.array_equal(outs[0], O.run_cf32(iq[:n])) and np.array_equal(outs[1], O.run_cf32(iq2[:n]))
        # the replay driver's bulk mode: reader thread + pinned blocks + the same call
        exe = os.path.join(ROOT, "build", "p25fe_replay")
        src, out = tmp_path / "cap.cf32", tmp_path / "dib.out"
        iq.tofile(src)
        r = subprocess.run([exe, "-W", "256k", "cf32", str(src), str(out)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        assert np.array_equal(np.fromfile(out, dtype=np.uint8), ref) and "windows" in r.stderr


@pytest.mark.timeout(300)
def test_bench_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE: the way the driver runs `--gpus 1`): bench.py starts the two
    ranks itself as children (torch.distributed.run), relays rank 0's line.  The step is the C ABI's (p25fe_shard_step through
    p25rx_amd/rccl.py); here with the shared-memory test hook, two ranks on this one GPU."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["P25FE_BENCH_HOST_STAGED"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--seconds", "30"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "launching 2 ranks" in out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "libp25fe_rccl" not in d["config"]["sharding"]     # (the hook says so)
    assert "TEST HOOK" in d["config"]["sharding"]
    assert d["config"]["parity_gate"].endswith("True") and d["config"]["gather_gate"].endswith("True")


def _rccl_cabi_world1_worker(q):
    import torch
    from p25rx_amd import c4fm, rccl
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import FrontEnd, parse_results
    torch.cuda.set_device(0)
    iq = c4fm.synth(2.0, seed=78, snr_db=22.0, frame_dibits=400)[0]
    n = len(iq) // 8 * 8
    fe = FrontEnd()
    ss = rccl.ShardStep(fe, 0, 1, n, rccl.unique_id())              # a ONE-rank RCCL communicator: self send / recv, all-gather of one
    halo = fe.shard_halo()
    buf = torch.zeros((halo + n, 2), dtype=torch.float32, device="cuda")
    buf[halo:] = torch.from_numpy(iq[:n].view(np.float32).reshape(-1, 2)).cuda()
    result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    dibits = torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device="cuda")
    ref, rres = FrontEnd().run_dev(buf[halo:])
    nref = int(parse_results(rres)[0]["n_dibits"])
    oks = []
    ss.comm_timing(1)                                               # events around the exchanges on every step
    for gather in ("root_exact", "root", "all", "root_exact"):
        for _ in range(3):
            ss.step(buf, dibits, result, gather=gather)
        torch.cuda.synchronize()
        off = ss.offsets()
        st = ss.stream(torch, "cuda", int(off[-1]))
        oks.append(int(off[-1]) == nref and bool(torch.equal(st, ref[0, :nref])) and ss.gather_ran() == gather)
    cm = ss.comm_ms()
    # (the one-rank communicator sends its dibits to itself: the sized send / recv of the exact mode runs here)
    oks.append(cm["steps"] == 12 and cm["halo_send_recv"] > 0 and cm["summary_all_gather"] > 0 and cm["dibit_gather"] > 0)
    ss.comm_timing(16)
    for _ in range(20):
        ss.step(buf, dibits, result, gather="root")
    torch.cuda.synchronize()
    oks.append(ss.comm_ms()["steps"] in (1, 2))                     # steps 12 .. 31: the 16th (and possibly the 32nd) carry events
    ss.close()
    q.put(oks)


@pytest.mark.timeout(300)
def test_c_abi_shard_step_through_ctypes_rccl_world1():
    """p25rx_amd/rccl.py (what bench.py --gpus N drives): p25fe_shard_step over a one-rank RCCL communicator, every gather
    mode -- exact bytes at their offsets, rows + compaction, all-gather -- gives the single-pass stream"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_cabi_world1_worker, args=(q,))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert res == [True, True, True, True, True, True]


def test_streams_share_queue_probe():
    """p25fe_streams_share_queue: a stream shares with itself; HIP gives the streams of one priority level four hardware queues, so of
    six normal-priority streams some pair shares one; a high-priority stream never shares with a normal-priority one.  (What
    p25fe_shard_step uses to keep its side stream off the caller's queue.)"""
    import torch
    from p25rx_amd.frontend import FrontEnd
    fe = FrontEnd()
    ss = [torch.cuda.Stream() for _ in range(6)]
    assert fe.streams_share_queue(ss[0], ss[0])
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6) if fe.streams_share_queue(ss[i], ss[j])]
    assert pairs, "six streams on four queues and no pair shares one?"
    assert len(pairs) <= 6
    hi = torch.cuda.Stream(priority=-1)
    assert not any(fe.streams_share_queue(hi, x) for x in ss)
    fe.streams_share_queue(None, ss[0])                              # the NULL stream is a legal argument


def _rccl_cabi_pipelined_worker(q):
    import torch
    from p25rx_amd import c4fm, rccl
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import FrontEnd, parse_results
    torch.cuda.set_device(0)
    oks = []
    caps = [c4fm.synth(2.0, seed=78, snr_db=22.0, frame_dibits=400)[0], c4fm.synth(2.0, seed=79, snr_db=22.0, frame_dibits=900)[0]]
    n = min(len(c) for c in caps) // 8 * 8
    # a one-rank RCCL communicator; no communicator at all; the tracking clock (pass 2 = resolve kernel + re-scan + slicer); and a rank whose
    # ncclCommSplit "failed" (P25FE_SHARD_SPLIT_FAIL): the layout every rank falls back to when ANY rank lacks a split -- one communicator,
    # the step's own order on the receive stream -- which is what a real N-GPU box runs the day a split fails
    for comm_id, clock, split_fail in ((rccl.unique_id(), 0, False), (None, 0, False), (rccl.unique_id(), 1, False), (rccl.unique_id(), 0, True)):
        fe = FrontEnd(symbol_clock=clock)
        if split_fail:
            os.environ["P25FE_SHARD_SPLIT_FAIL"] = "all"
        ss = rccl.ShardStep(fe, 0, 1, n, comm_id)
        os.environ.pop("P25FE_SHARD_SPLIT_FAIL", None)
        ss.prepare()
        inf = ss.info()
        want = (0, 0, 2) if comm_id is None else ((1, 1, 1) if split_fail else (1, 3, 2))
        oks.append((inf["rccl_ranks"], inf["comms"], inf["pipe_layout"]) == want and inf["world"] == 1 and not inf["staged"]
                   and not inf["broken"] and len(inf["pci_bus_id"]) >= 7 and inf["device"] == 0)
        halo = fe.shard_halo()
        bufs, refs = [], []
        for iq in caps:
            b = torch.zeros((halo + n, 2), dtype=torch.float32, device="cuda")
            b[halo:] = torch.from_numpy(iq[:n].view(np.float32).reshape(-1, 2)).cuda()
            bufs.append(b)
            ref, rres = FrontEnd(symbol_clock=clock).run_dev(b[halo:])
            refs.append(ref[0, :int(parse_results(rres)[0]["n_dibits"])].clone())
        order = [0, 1, 1, 0, 1, 0, 0, 1, 0]
        outs = [(torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device="cuda"),
                 torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")) for _ in order]
        for gather in ("root", "all", "root_exact", "none"):
            for k, w in enumerate(order):                            # all enqueued back to back: step k + 1's K1 beside step k's chain
                ss.step(bufs[w], outs[k][0], outs[k][1], gather=gather, pipelined=True)
            ss.join()
            torch.cuda.synchronize()
            ok = True
            for k, w in enumerate(order):
                nd = int(parse_results(outs[k][1])[0]["n_dibits"])
                ok = ok and nd == len(refs[w]) and bool(torch.equal(outs[k][0][0, :nd], refs[w]))
            if gather != "none":
                off = ss.offsets()
                ok = ok and int(off[-1]) == len(refs[order[-1]]) and bool(torch.equal(ss.stream(torch, "cuda", int(off[-1])), refs[order[-1]]))
            oks.append(ok)
        # a plain step straight behind a pipelined one (no join): it waits for the receive stream by itself
        ss.step(bufs[1], outs[0][0], outs[0][1], gather="root", pipelined=True)
        ss.step(bufs[0], outs[1][0], outs[1][1], gather="root")
        torch.cuda.synchronize()
        off = ss.offsets()
        n1 = int(parse_results(outs[0][1])[0]["n_dibits"])
        oks.append(n1 == len(refs[1]) and bool(torch.equal(outs[0][0][0, :n1], refs[1])) and int(off[-1]) == len(refs[0])
                   and bool(torch.equal(ss.stream(torch, "cuda", int(off[-1])), refs[0])))
        ss.close()
    q.put(oks)


@pytest.mark.timeout(300)
def test_c_abi_shard_step_pipelined():
    """p25fe_shard_step_pipelined: nine steps over two alternating captures enqueued back to back (every gather mode; with a one-rank
    RCCL communicator, without one, and on the one-communicator fallback layout of a failed ncclCommSplit) leave, per step, the
    single-pass dibits of THAT step's capture, and the ordered stream of the last one; a plain step behind a pipelined one needs no
    join; p25fe_shard_info says which layout ran."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_cabi_pipelined_worker, args=(q,))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert res == [True] * 24


@pytest.mark.timeout(300)
def test_bench_two_ranks_channel_blocks_host_staged():
    """config 4 over N GPUs (ChannelShard): bench.py --workload channels with two ranks on one GPU.  5 channels -> blocks
    of 3 and 2; every rank decodes its block with the real kernels, the gathered summary table is complete."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, P25FE_BENCH_HOST_STAGED="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "channels", "--channels", "5",
                          "--seconds", "4"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["n_gpus"] == 2 and "5 independent channels" in d["config"]["workload"]
    assert d["config"]["parity_gate"].endswith("True") and d["config"]["table_gate"].endswith("True")


def _rccl_world1_worker(port, q):
    import torch
    import torch.distributed as dist
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.sharding import TorchComm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        comm = TorchComm(dist, 0, 1)
        item = RESULT_DTYPE.itemsize
        summ = torch.arange(item, dtype=torch.uint8, device="cuda").reshape(1, item)
        out = torch.zeros((1, item), dtype=torch.uint8, device="cuda")
        comm.all_gather(out, summ)                                  # all_gather_into_tensor on uint8 views: RCCL
        dib = (torch.arange(4096, device="cuda") % 4).to(torch.uint8).reshape(1, -1)
        gathered = torch.zeros((1, 4096), dtype=torch.uint8, device="cuda")
        comm.all_gather(gathered, dib)
        root = torch.zeros((1, 4096), dtype=torch.uint8, device="cuda")
        comm.gather_to_root(root, dib)                              # world 1: the root's own row
        works = comm.halo_start(dib, dib)                           # no neighbours: nothing to do
        comm.halo_wait(works)
        # the point-to-point calls of halo_start / gather_to_root (batched P2POp isend + irecv), looped back to this rank:
        # a halo-sized slice of a cf32 capture and a dibit row
        cap = torch.arange(2 * 40000, dtype=torch.float32, device="cuda").reshape(-1, 2)
        halo_in = torch.zeros((1728, 2), dtype=torch.float32, device="cuda")
        row = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, cap[-1728:], 0), dist.P2POp(dist.irecv, halo_in, 0)]):
            w.wait()
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, dib.reshape(-1), 0), dist.P2POp(dist.irecv, row, 0)]):
            w.wait()
        dist.barrier()
        torch.cuda.synchronize()
        q.put((bool(torch.equal(out, summ)), bool(torch.equal(gathered, dib)), bool(torch.equal(root, dib)), len(works),
               bool(torch.equal(halo_in, cap[-1728:])), bool(torch.equal(row, dib.reshape(-1)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_backend_world1_collectives():
    """The collectives of the multi-GPU step on the REAL backend (nccl = RCCL), as far as one GPU allows: process group
    of one rank, all_gather_into_tensor of the shard summary and of a dibit buffer as uint8 device tensors, the root's
    own row of the point-to-point gather, barrier, and the batched isend / irecv of the halo exchange and of the dibit
    gather looped back to the same rank (RCCL allows send / recv to self inside a group).  What one GPU cannot show is a
    transfer over xGMI: the two-rank logic is covered by gloo."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert res == (True, True, True, 0, True, True)


def _rccl_world1_step_worker(port, q):
    import torch
    import torch.distributed as dist
    from p25rx_amd import c4fm
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd.sharding import TimeShard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))    # as bench.py does for N > 1
    try:
        iq = c4fm.synth(2.0, seed=77, snr_db=22.0, frame_dibits=400)[0]
        n = len(iq) // 8 * 8
        fe = FrontEnd()
        ts = TimeShard(fe, 0, 1, n, dist)
        ts.always_comm = True
        ts.setup_device(torch, "cuda")
        buf = ts.alloc(torch, "cuda", torch.float32)
        buf[ts.halo:] = torch.from_numpy(iq[:n].view(np.float32).reshape(-1, 2)).cuda()
        result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        summ_all = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        dibits = torch.zeros((1, ts.dibit_cap), dtype=torch.uint8, device="cuda")
        oks = []
        ref, rres = FrontEnd().run_dev(buf[ts.halo:])
        nref = int(parse_results(rres)[0]["n_dibits"])
        for gather in ("root", "all", "root"):
            ts.d_stream.zero_()
            off = ts.step_device(buf, result, summ_all, dibits, gather=gather)     # no host synchronisation inside
            torch.cuda.synchronize()
            total = int(off[-1])
            oks.append(total == nref and bool(torch.equal(ts.d_stream[:total], ref[0, :nref])))
        q.put(oks)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_timeshard_step_device_rccl_world1():
    """The multi-GPU step with its collectives on the real backend (RCCL), one rank: pass 1 main / finish, the summary
    all_gather, the device resolve, pass 2, the dibit gather (to the root, and as all_gather) and the compaction run as
    they do for N > 1 -- enqueued without a host synchronisation between library kernels and RCCL operations -- and the
    gathered stream equals the single-pass dibits."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_step_worker, args=(port, q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert res == [True, True, True]


@pytest.mark.timeout(600)
def test_c_abi_shard_step_rccl_and_two_processes(tmp_path):
    """include/p25fe_rccl.h through the C++ launcher (no Python, no torch in the ranks): the N > 1 step -- halo exchange
    beside K1, summary all-gather, device resolve, pass 2, dibit gather to rank 0, compaction -- (a) over RCCL in a
    one-rank communicator on this GPU, (b) as 2, 3 and 8 PROCESSES that share the GPU with the exchanges staged through shared
    memory (test hook).  The ordered stream rank 0 writes equals the single-pass dibits byte for byte."""
    import json
    from oracle import oracle as O
    from p25rx_amd import c4fm
    exe = os.path.join(ROOT, "build", "p25fe_shards")
    iq, _, _ = c4fm.synth(2.0, seed=91, snr_db=24.0, frame_dibits=700)
    iq = iq[:480000]
    ref = O.run_cf32(iq)
    src = tmp_path / "cap.cf32"
    iq.tofile(src)
    for args in (["-n", "1", "-t", "1"], ["-n", "1", "-g", "exact", "-t", "1"], ["-n", "1"], ["-n", "2", "--shm"], ["-n", "3", "--shm"],
                 ["-n", "8", "--shm"], ["-n", "2", "--shm", "-g", "exact"], ["-n", "3", "--shm", "-g", "exact"],
                 ["-n", "8", "--shm", "-g", "exact"]):
        out = tmp_path / ("dib_" + "_".join(a.strip("-") for a in args))
        r = subprocess.run([exe] + args + ["-k", "3", str(src), str(out)], capture_output=True, timeout=280)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        rep = json.loads(r.stdout.decode().strip().splitlines()[-1])
        got = np.fromfile(out, dtype=np.uint8)
        assert rep["dibits"] == len(got) == len(ref), (args, rep)
        assert np.array_equal(got, ref), args
        # the mode that RAN (p25fe_shard_gather_ran), also through the shared-memory hook, which now executes the exact-bytes
        # gather's offset arithmetic with 2, 3 and 8 ranks
        assert rep["gather"] == ("exact" if "exact" in args else "rows"), (args, rep)
        if "--shm" not in args:
            assert rep["exchange"] == "RCCL"
            if "-t" in args:                                        # events on every step: three timed steps
                assert rep["comm_ms_per_step"]["steps_averaged"] == 3
                assert rep["comm_ms_per_step"]["halo"] > 0 and rep["comm_ms_per_step"]["summaries"] > 0
                assert rep["comm_ms_per_step"]["dibit_gather"] > 0   # the one-rank communicator sends to itself
            else:                                                   # library default: every 16th step (step 0 of the warm-up)
                assert rep["comm_ms_per_step"]["steps_averaged"] <= 1
    # the same with the tracking symbol clock on a capture whose sample clock is 150 ppm off: every shard's first detection
    # takes its period from the previous shard's anchor (the carry resolution with clocks, p25fe_shard_resolve_dev)
    iq2, _, _ = c4fm.synth(2.0, seed=92, snr_db=24.0, frame_dibits=700, clock_ppm=150.0)
    iq2 = iq2[:480000]
    bb2 = O.Demod().feed_cf32(iq2)
    ref2 = O.Recv(O.make_config(symbol_clock=1)).feed(bb2)[0]
    src2 = tmp_path / "cap_ppm.cf32"
    iq2.tofile(src2)
    for args in (["-n", "1", "-c", "1"], ["-n", "3", "-c", "1", "--shm"]):
        out = tmp_path / ("dibt_" + "_".join(a.strip("-") for a in args))
        r = subprocess.run([exe] + args + ["-k", "2", str(src2), str(out)], capture_output=True, timeout=280)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = np.fromfile(out, dtype=np.uint8)
        assert len(got) == len(ref2) and np.array_equal(got, ref2), args


@pytest.mark.timeout(120)
def test_cpp_replay_bulk_mode_fails_cleanly(tmp_path):
    """p25fe_replay -W (reader thread + pinned blocks + p25fe_run_host_windows): an output that cannot be written ends in a clean
    non-zero exit with a message -- the reader thread joined, no std::terminate / abort -- not in a silently short dibit file."""
    from p25rx_amd import c4fm
    exe = os.path.join(ROOT, "build", "p25fe_replay")
    src = tmp_path / "cap.cf32"
    c4fm.synth(1.0, seed=5, snr_db=25.0)[0].tofile(src)
    r = subprocess.run([exe, "-W", "256k", "cf32", str(src), "/dev/full"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 1 and "write error" in r.stderr, (r.returncode, r.stderr[-500:])
    r = subprocess.run([exe, "-W", "256k", "cf32", str(src), str(tmp_path / "no_such_dir" / "x")], capture_output=True, text=True, timeout=100)
    assert r.returncode == 1 and "unable to open" in r.stderr, (r.returncode, r.stderr[-500:])
