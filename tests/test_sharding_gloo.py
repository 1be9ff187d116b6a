"""world_size-2 CPU tests (gloo) of the multi-GPU orchestration in p25rx_amd/sharding.py.

The GPU kernels cannot run here, so the per-shard compute is a TEST DOUBLE built on the CPU oracle
(OracleShardFE below); what is under test is the product's N > 1 logic: the halo exchange (blocking and
split around pass 1), the all-gather of shard summaries, p25fe_shard_resolve (real C-ABI host function), the
resulting carry-in / dibit offsets, the all-gather + compaction of the dibit shards into one ordered stream --
through TimeShard.step and through TimeShard.step_device (the bench's N > 1 step) -- and ChannelShard's channel
blocks and summary gather.  The gathered stream must equal one oracle pass over the capture.
"""
import os
import socket

import numpy as np
import pytest

N_PER_RANK = 96000        # 0.4 s per rank
WORLD = 2


class OracleShardFE:
    """Test double with the shard_* surface of p25rx_amd.frontend.FrontEnd, computed by the oracle."""

    def __init__(self):
        from oracle import oracle as O
        from p25rx_amd import _lib
        self.O, self._lib = O, _lib
        self.W = O.load_spec()["sync_peak_w"]

    def shard_halo(self):
        return self._lib.load().p25fe_shard_halo()

    def shard_pass1(self, view, offset, n_hist, abs0, result=None):
        import torch
        from p25rx_amd.frontend import n_baseband
        O = self.O
        x = view.numpy().view(np.complex64).reshape(-1)
        pad = (abs0 - n_hist) % 5                                   # keep the absolute 5:1 grid (src/demod.rs:87-90)
        bb_all = O.Demod().feed_cf32(np.concatenate([np.zeros(pad, np.complex64), x]))
        nb = n_baseband(abs0, len(x) - offset)
        self.bb0 = n_baseband(0, abs0)
        self.bb = bb_all[len(bb_all) - nb:]
        hist = bb_all[max(0, len(bb_all) - nb - 256):len(bb_all) - nb]
        r = O.Recv()
        r.feed(hist)
        r.resync()                                                  # events decided before the shard belong to the previous one
        dib, spos, sdib = r.feed(self.bb)
        base = self.bb0 - len(hist)                                 # absolute index of hist[0]
        self.own = (dib, spos + base)
        st = r.state()
        summ = np.zeros(1, dtype=self._lib.RESULT_DTYPE)
        summ["n_baseband"] = nb
        summ["first_event"] = summ["carry_end"] = -1
        if len(spos):
            summ["first_event"] = spos[0] + base + self.W
            summ["carry_end"] = summ["first_event"] + 1
            summ["n_dibits_after_first"] = len(dib)
            summ["anchor_out"] = (st["s"] + base, st["hi"], st["mid"], st["lo"], 1, 10, 1)
        out = torch.from_numpy(np.frombuffer(summ.tobytes(), dtype=np.uint8).copy()).view(1, -1)
        if result is not None:
            result.copy_(out)
            return result
        return out

    def shard_pass1_main(self, view, offset, n_hist, abs0):
        self._main = (offset, n_hist, abs0)                        # the real one launches K1 for everything but the head

    def shard_pass1_finish(self, view, offset, n_hist, abs0, result=None):
        assert self._main == (offset, n_hist, abs0)
        return self.shard_pass1(view, offset, n_hist, abs0, result=result)

    def shard_compact_dev(self, gathered, offsets, out):
        off = offsets.numpy()
        for r in range(gathered.shape[0]):
            out[int(off[r]):int(off[r + 1])] = gathered[r, :int(off[r + 1] - off[r])]
        return out

    def shard_resolve(self, summaries, bb0, bbn):
        import ctypes as C
        L = self._lib.load()
        summaries = np.ascontiguousarray(summaries, dtype=self._lib.RESULT_DTYPE)
        bb0 = np.ascontiguousarray(bb0, dtype=np.uint64)
        bbn = np.ascontiguousarray(bbn, dtype=np.uint64)
        n = len(summaries)
        anc = np.zeros(n, dtype=self._lib.ANCHOR_DTYPE)
        off = np.zeros(n + 1, dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.p25fe_shard_resolve(p(summaries), p(bb0), p(bbn), n, 0, p(anc), p(off)) == 0
        return anc, off

    def shard_resolve_dev(self, summ_all, d_bb0, d_bbn, anchors=None, offsets=None):
        """Stand-in for the one-thread device kernel: the same C-ABI host function on CPU tensors."""
        import torch
        summ = np.frombuffer(summ_all.numpy().tobytes(), dtype=self._lib.RESULT_DTYPE)
        anc, off = self.shard_resolve(summ, d_bb0.numpy().astype(np.uint64), d_bbn.numpy().astype(np.uint64))
        a_t = torch.from_numpy(np.frombuffer(anc.tobytes(), dtype=np.uint8).copy()).view(len(anc), -1)
        return a_t, torch.from_numpy(off.astype(np.int64))

    def shard_pass2(self, anchor_in, n_bb, device, result=None, dibits=None):
        if not isinstance(anchor_in, np.ndarray):                # uint8 tensor holding one p25fe_anchor_t
            anchor_in = np.frombuffer(anchor_in.numpy().tobytes(), dtype=self._lib.ANCHOR_DTYPE)
        a = anchor_in[0]
        dib, spos = self.own
        pre = np.zeros(0, np.uint8)
        if a["valid"]:
            hi_idx = (spos[0] + self.W) if len(spos) else self.bb0 + len(self.bb) - 1   # instants <= first decision index
            n = np.arange(self.bb0, hi_idx + 1)
            n = n[(n > a["s"]) & ((n - a["s"]) % 10 == 0)]
            v = self.bb[n - self.bb0]
            pre = np.where(v >= a["hi"], 1, np.where(v >= a["mid"], 0, np.where(v >= a["lo"], 2, 3))).astype(np.uint8)
        self.out = np.concatenate([pre, dib])
        if dibits is not None:
            import torch
            dibits[0, :len(self.out)] = torch.from_numpy(self.out)
        return None, None


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from p25rx_amd import c4fm
    from p25rx_amd.sharding import TimeShard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        iq, _, _ = c4fm.synth(world * N_PER_RANK / 240000.0, seed=77, snr_db=22.0, frame_dibits=700)
        fe = OracleShardFE()
        ts = TimeShard(fe, rank, world, N_PER_RANK, dist)
        buf = ts.alloc(torch, "cpu", torch.float32)
        mine = iq[rank * N_PER_RANK:(rank + 1) * N_PER_RANK]
        buf[ts.halo:] = torch.from_numpy(mine.view(np.float32).reshape(-1, 2))
        result = torch.zeros((1, fe._lib.RESULT_DTYPE.itemsize), dtype=torch.uint8)
        summ_all = torch.zeros((world, fe._lib.RESULT_DTYPE.itemsize), dtype=torch.uint8)
        off, summ, offs = ts.step(buf, result, summ_all, None)
        counts, total = ts.gather_counts(offs)
        assert len(counts) == world and total == int(offs[-1]) and counts[rank] == len(fe.out)
        if rank > 0:                                            # halo really is the left neighbour's tail
            left = iq[rank * N_PER_RANK - ts.halo:rank * N_PER_RANK]
            assert np.array_equal(buf[:ts.halo].numpy().view(np.complex64).reshape(-1), left)
        out_host = fe.out
        # the bench's variant of the step: no host synchronisation, carry resolved from the all-gathered tensor
        ts.setup_device(torch, "cpu")
        buf[:ts.halo].zero_()
        dibits = torch.zeros((1, ts.dibit_cap), dtype=torch.uint8)
        d_off = ts.step_device(buf, result, summ_all, dibits)       # halo overlapped with pass 1, dibit gather + compaction
        if rank > 0:
            assert np.array_equal(buf[:ts.halo].numpy().view(np.complex64).reshape(-1), left)
        assert np.array_equal(fe.out, out_host)
        assert int(d_off[rank]) == off and int(d_off[-1]) == total
        stream = ts.d_stream[:total].numpy().copy() if rank == 0 else None    # gathered to the root (the consumer)
        d_off2 = ts.step_device(buf, result, summ_all, dibits, gather="all")    # all-gather form: the stream on EVERY rank
        assert d_off2.tolist() == d_off.tolist()
        stream_all = ts.d_stream[:total].numpy().copy()
        assert stream is None or np.array_equal(stream, stream_all)
        stream = stream_all
        # channel shards (config 4 over N GPUs): blocks partition the batch; the summary gather keeps channel order
        from p25rx_amd.sharding import ChannelShard
        cs = ChannelShard(rank, world, 5, dist)
        mine_res = torch.arange(cs.c0, cs.c1, dtype=torch.uint8).view(-1, 1).repeat(1, 8)
        allres = cs.gather_results(torch, mine_res)
        assert allres.shape == (5, 8) and allres[:, 0].tolist() == [0, 1, 2, 3, 4]
        q.put((rank, off, fe.out, O.run_cf32(iq) if rank == 0 else None, stream))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_time_shards_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=150) for _ in range(WORLD)])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    ref = got[0][3]
    outs = [g[2] for g in got]
    assert got[0][1] == 0 and got[1][1] == len(outs[0])         # dibit offsets from p25fe_shard_resolve
    assert np.array_equal(np.concatenate(outs), ref[:sum(len(o) for o in outs)])
    assert abs(len(ref) - sum(len(o) for o in outs)) <= 1
    for g in got:                                                # the gathered stream is the same on both ranks
        assert np.array_equal(g[4], np.concatenate(outs))


def test_channel_blocks_partition():
    from p25rx_amd.sharding import ChannelShard
    for world in (1, 2, 3, 8):
        for n in (1, 7, 8, 256, 257):
            blocks = [ChannelShard.block(r, world, n) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_bench_plain_invocation_reaches_the_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must not die on an assert (round 3 did): bench.py starts
    the two ranks itself, as children, through torch.distributed.run.  Without a GPU the ranks get as far as the rendezvous
    (gloo, world size 2) and then refuse to run -- there is no CPU fallback -- which is what this CPU test can see; the same
    command completes on the GPU box (tests/test_gpu_host.py)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_host.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["P25FE_BENCH_HOST_STAGED"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--seconds", "1",
                          "--no-extra", "--no-cpu"], env=env, capture_output=True, text=True, timeout=240)
    assert "launching 2 ranks" in out.stderr and "torch.distributed.run" in out.stderr
    assert out.returncode != 0                                           # no GPU: the ranks fail loudly ...
    assert out.stderr.count("No HIP GPUs are available") >= 2           # ... both of them, after the launcher started them
    assert "--gpus must equal WORLD_SIZE" not in out.stderr
    # ... and rank 0 leaves ONE line that says what failed and where, with the contract's keys (a failed N > 1 run is a record, not silence)
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 0.0 and d["n_gpus"] == 2 and "No HIP GPUs" in d["error"] and d["stage"] and d["metric"].startswith("IQ Msamples/s")


def test_bench_watchdog_ends_a_rank_stuck_in_the_rendezvous():
    """A rank whose peers never arrive (a dead GPU, a wedged RCCL bootstrap) must not hang until somebody else's timeout: after
    P25FE_BENCH_WATCHDOG_S rank 0 prints the failure line -- with the stage it was stuck in -- and the process ends with code 5."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               P25FE_BENCH_WATCHDOG_S="20", P25FE_BENCH_HOST_STAGED="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--seconds", "1",
                          "--no-extra", "--no-cpu"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 5, (out.returncode, out.stderr[-500:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["error"].startswith("watchdog") and "rendezvous" in d["stage"] and d["value"] == 0.0


def test_bench_rank0_leaves_its_line_when_the_launcher_terminates_it():
    """torch.distributed.run ends the surviving ranks with SIGTERM when another rank fails first; rank 0 -- the rank whose line the driver
    reads -- still prints the failure line (exit code 6)."""
    import json
    import signal
    import socket
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               P25FE_BENCH_WATCHDOG_S="200", P25FE_BENCH_HOST_STAGED="1")
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--seconds", "1",
                          "--no-extra", "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8.0)                                                  # (the handler is installed before torch is imported)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=120)
    assert p.returncode == 6, (p.returncode, err[-400:])
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert d["error"].startswith("terminated by the launcher") and d["value"] == 0.0 and d["n_gpus"] == 2
