"""Sharding across the GPUs of one node (BASELINE.json configs 4 and 5).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

TimeShard -- one long capture cut into contiguous time ranges, rank r owns [r*n, (r+1)*n).  Exchange steps:

  1. filter-state overlap: the last `halo` IQ samples of rank r go to rank r+1 (point-to-point; on MI355X one xGMI
     link, 2 048 samples = 16 KB of cf32).  The receive is hidden behind K1: everything that does not touch the halo is launched first
     (p25fe_shard_pass1_main), the shard's head after the halo has arrived (p25fe_shard_pass1_finish);
  2. symbol-timing carry: every rank's 96-byte shard summary (p25fe_result_t) is all-gathered; a one-thread kernel resolves every
     shard's carry-in anchor and dibit offset (p25fe_shard_resolve_dev) -- no host synchronisation;
  3. the reduced dibit stream: after pass 2 the shards' dibit buffers (n/50 bytes each) are GATHERED TO THE ROOT rank
     (point-to-point, every rank has a direct xGMI link to it) and compacted there into ONE ordered stream
     (p25fe_shard_compact_dev) -- what RecvTask feeds into MessageReceiver, src/recv.rs:148-150.  For the one-hour
     capture of config 5 that is 17.3 MB in total.  (gather="all" all-gathers instead: the stream on every rank, at N
     times the traffic.)

ChannelShard -- a batch of independent channels (config 4): rank r takes a contiguous block of channels, no
communication on the data path at all (one tuner = one channel in the reference, src/sdr.rs:64-65); the per-channel
summaries can be all-gathered for a consumer that wants the whole batch's lock / count table.

The reference has no distributed code; the serial state it carries across chunks
(src/demod.rs:25-40, MessageReceiver's lock) is what steps 1 and 2 hand over.

`comm` objects isolate the collectives: TorchComm issues them on device tensors (RCCL / gloo); HostStagedComm stages
through the CPU so that the WHOLE step, library calls included, can be exercised by several processes that share one GPU
(tests) -- the product path is TorchComm.
"""
import numpy as np

from ._lib import RESULT_DTYPE


class TorchComm:
    """Collectives on the tensors as they are (device tensors with RCCL, CPU tensors with gloo)."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def halo_start(self, send_t, recv_t):
        """Start the neighbour exchange; returns a handle for halo_wait."""
        d, ops = self.dist, []
        if self.rank + 1 < self.world:
            ops.append(d.P2POp(d.isend, send_t, self.rank + 1))
        if self.rank > 0:
            ops.append(d.P2POp(d.irecv, recv_t, self.rank - 1))
        return d.batch_isend_irecv(ops) if ops else []

    def halo_wait(self, works):
        for w in works:
            w.wait()                    # device tensors: the current stream waits, the host does not

    def all_gather(self, out_t, in_t):
        self.dist.all_gather_into_tensor(out_t.view(-1), in_t.reshape(-1))

    def gather_to_root(self, out_t, in_t, root=0):
        """Row r of out_t (root only) <- rank r's in_t.  Point-to-point: on MI355X every rank has a direct xGMI link to
        the root, so the 7 shards arrive in parallel instead of travelling a ring (an all-gather moves N times the
        bytes the one consumer needs)."""
        d, ops = self.dist, []
        if self.rank == root:
            out_t[root].copy_(in_t.reshape(-1))
            for r in range(self.world):
                if r != root:
                    ops.append(d.P2POp(d.irecv, out_t[r], r))
        else:
            ops.append(d.P2POp(d.isend, in_t.reshape(-1), root))
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()


class HostStagedComm(TorchComm):
    """The same exchanges through CPU copies (gloo): several ranks can then share ONE GPU."""

    def halo_start(self, send_t, recv_t):
        d, ops = self.dist, []
        self._recv_dev, self._recv_cpu = recv_t, None
        if self.rank + 1 < self.world:
            self._send_cpu = send_t.cpu().contiguous()
            ops.append(d.P2POp(d.isend, self._send_cpu, self.rank + 1))
        if self.rank > 0:
            self._recv_cpu = recv_t.cpu().contiguous()
            ops.append(d.P2POp(d.irecv, self._recv_cpu, self.rank - 1))
        return d.batch_isend_irecv(ops) if ops else []

    def halo_wait(self, works):
        for w in works:
            w.wait()
        if self._recv_cpu is not None:
            self._recv_dev.copy_(self._recv_cpu)

    def all_gather(self, out_t, in_t):
        o = out_t.cpu().contiguous()
        self.dist.all_gather_into_tensor(o.view(-1), in_t.cpu().contiguous().reshape(-1))
        out_t.copy_(o)

    def gather_to_root(self, out_t, in_t, root=0):
        d, ops = self.dist, []
        if self.rank == root:
            o = out_t.cpu().contiguous()
            o[root].copy_(in_t.reshape(-1).cpu())
            for r in range(self.world):
                if r != root:
                    ops.append(d.P2POp(d.irecv, o[r], r))
        else:
            ops.append(d.P2POp(d.isend, in_t.reshape(-1).cpu().contiguous(), root))
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        if self.rank == root:
            out_t.copy_(o)


class TimeShard:
    """Per-rank driver of a time-sharded capture.  `fe` is a one-channel FrontEnd (or a test double with the same
    shard_* methods); rank r owns samples [r * n_per_rank, (r + 1) * n_per_rank)."""

    def __init__(self, fe, rank, world, n_per_rank, dist=None, comm=None):
        if getattr(fe, "C", 1) != 1:
            raise ValueError("TimeShard drives a one-channel handle (shard summaries and the resolve are per channel)")
        if n_per_rank % 8:
            raise ValueError("n_per_rank must be a multiple of 8 samples (shard cut points stay 16-byte aligned)")
        self.fe, self.rank, self.world, self.n, self.dist = fe, rank, world, n_per_rank, dist
        self.comm = comm if comm is not None else (TorchComm(dist, rank, world) if dist is not None else None)
        self.always_comm = False         # tests: take the collective path at world = 1 too (RCCL on a one-GPU box)
        self.comm_events = None          # bench: list that receives (phase, start event, end event) per exchange of a step
        self.halo = int(fe.shard_halo())
        self.abs0 = rank * n_per_rank
        from .frontend import n_baseband as _nb_default
        n_baseband = getattr(fe, "n_baseband", _nb_default)      # (the handle's own decimator phase; test doubles have none)
        self.bb0 = [n_baseband(0, r * n_per_rank) for r in range(world)]
        self.bbn = [n_baseband(r * n_per_rank, n_per_rank) for r in range(world)]
        # per-shard dibit buffer (gather granule): n / 10 plus proportional slack -- a receiver that re-anchors on every sync
        # word follows the TRANSMITTER's symbol clock (200 ppm here; the kernels stop storing at the row's end and the count
        # stays exact, so a larger offset is detected, not overrun)
        self.dibit_cap = (max(self.bbn) // 10 + max(self.bbn) // 50000 + 64 + 15) // 16 * 16

    def alloc(self, torch, device, dtype):
        """[halo | owned] buffer; the halo part is filled by the neighbour exchange."""
        return torch.zeros((self.halo + self.n, 2), dtype=dtype, device=device)

    # ---- host-resolved form (tests, small captures): synchronises once per step ---------------------------------
    def exchange_halo(self, buf):
        """Step 1, blocking form: last `halo` samples of my range -> rank+1; rank-1's arrive in buf[:halo]."""
        if self.world > 1:
            self.comm.halo_wait(self.comm.halo_start(buf[self.n:], buf[:self.halo]))

    def pass1(self, buf, result):
        h = self.halo if self.rank > 0 else 0
        return self.fe.shard_pass1(buf[self.halo - h:], offset=h, n_hist=h, abs0=self.abs0, result=result)

    def exchange_summaries(self, result, summ_all):
        """Step 2: all-gather the per-rank summaries (one p25fe_result_t each) and resolve the carry on the host."""
        if self.world > 1:
            self.comm.all_gather(summ_all, result)
            raw = summ_all.cpu().numpy().tobytes()
        else:
            raw = result.cpu().numpy().tobytes()
        summ = np.frombuffer(raw, dtype=RESULT_DTYPE)
        anchors, offsets = self.fe.shard_resolve(summ, self.bb0, self.bbn)
        return summ, anchors, offsets

    def pass2(self, anchors, device, result, dibits):
        return self.fe.shard_pass2(anchors[self.rank:self.rank + 1], self.bbn[self.rank], device, result=result,
                                   dibits=dibits)

    def step(self, buf, result, summ_all, dibits):
        """One pass of the hot path over my shard; returns (dibit_offset_of_my_shard, summaries, all offsets)."""
        self.exchange_halo(buf)
        self.pass1(buf, result)
        summ, anchors, offsets = self.exchange_summaries(result, summ_all)
        self.pass2(anchors, buf.device, result, dibits)
        return int(offsets[self.rank]), summ, offsets

    def gather_counts(self, offsets):
        """Global dibit layout from the resolved offsets (n_shards + 1 entries): per-shard counts and the total."""
        off = np.asarray(offsets, dtype=np.int64)
        return np.diff(off), int(off[-1])

    # ---- device-resolved form (the bench): no host synchronisation anywhere in a step -----------------------------
    def setup_device(self, torch, device):
        self._torch = torch
        self.d_bb0 = torch.tensor(self.bb0, dtype=torch.int64, device=device)
        self.d_bbn = torch.tensor(self.bbn, dtype=torch.int64, device=device)
        self.d_anchors = self.d_offsets = None
        self.d_gathered = torch.empty((self.world, self.dibit_cap), dtype=torch.uint8, device=device)
        self.d_stream = torch.empty(self.world * self.dibit_cap, dtype=torch.uint8, device=device)

    def _mark(self, phase=None, start=None):
        """bench only (comm_events is a list): an event on the current stream; with `phase`, files (phase, start, end)."""
        if self.comm_events is None or not getattr(self, "_torch", None) or not self._torch.cuda.is_available():
            return None
        ev = self._torch.cuda.Event(enable_timing=True)
        ev.record()
        if phase is not None and start is not None:
            self.comm_events.append((phase, start, ev))
        return ev

    def step_device(self, buf, result, summ_all, dibits, gather="root"):
        """halo exchange (overlapped with K1) -> pass 1 -> all_gather of summaries -> k_shard_resolve -> pass 2 ->
        gather of the dibit shards -> compaction, all enqueued on the current stream.  `dibits` must be a
        [1, dibit_cap] buffer.  Returns the device tensor of n_shards + 1 dibit offsets (read it after the timed
        region); the ordered stream of the whole capture is self.d_stream[:offsets[-1]] on rank 0 (gather="root") or on
        every rank (gather="all"); gather=None leaves the stream sharded."""
        h = self.halo if self.rank > 0 else 0
        view = buf[self.halo - h:]
        multi = self.world > 1 or (self.comm is not None and self.always_comm)
        mark = self._mark
        if multi:
            works = self.comm.halo_start(buf[self.n:], buf[:self.halo])
            self.fe.shard_pass1_main(view, offset=h, n_hist=h, abs0=self.abs0)     # needs no halo
            e = mark()
            self.comm.halo_wait(works)
            mark("halo_wait_after_k1", e)
            self.fe.shard_pass1_finish(view, offset=h, n_hist=h, abs0=self.abs0, result=result)
            e = mark()
            self.comm.all_gather(summ_all, result)
            mark("summaries_all_gather", e)
        else:
            self.fe.shard_pass1(view, offset=h, n_hist=h, abs0=self.abs0, result=result)
            summ_all.copy_(result)
        self.d_anchors, self.d_offsets = self.fe.shard_resolve_dev(summ_all, self.d_bb0, self.d_bbn, self.d_anchors,
                                                                   self.d_offsets)
        self.fe.shard_pass2(self.d_anchors[self.rank:self.rank + 1], self.bbn[self.rank], buf.device, result=result,
                            dibits=dibits)
        if gather:
            e = mark()
            if not multi:
                self.d_gathered.copy_(dibits)
            elif gather == "all":
                self.comm.all_gather(self.d_gathered, dibits)
            else:
                self.comm.gather_to_root(self.d_gathered, dibits, root=0)
            if multi:
                mark("dibit_gather", e)
            if gather == "all" or self.rank == 0:
                self.fe.shard_compact_dev(self.d_gathered, self.d_offsets, self.d_stream)
        return self.d_offsets


class ChannelShard:
    """Config 4 over N GPUs: rank r owns the contiguous channel block [c0, c1) of a batch of n_channels; the data path
    has no collective.  make_frontend(n_local) must return a FrontEnd for that many channels."""

    def __init__(self, rank, world, n_channels, dist=None, comm=None):
        self.rank, self.world, self.n_channels = rank, world, n_channels
        self.c0, self.c1 = self.block(rank, world, n_channels)
        self.comm = comm if comm is not None else (TorchComm(dist, rank, world) if dist is not None else None)

    @staticmethod
    def block(rank, world, n_channels):
        """Balanced contiguous blocks: the first n_channels % world ranks take one channel more."""
        q, r = divmod(n_channels, world)
        c0 = rank * q + min(rank, r)
        return c0, c0 + q + (1 if rank < r else 0)

    @property
    def n_local(self):
        return self.c1 - self.c0

    def step(self, fe, iq_local, dibits=None, result=None):
        """One pass over my channels ([n_local, n, 2] device tensor): exactly FrontEnd.run_dev, nothing else."""
        return fe.run_dev(iq_local, dibits=dibits, result=result)

    def gather_results(self, torch, result_local):
        """Per-channel summaries of the whole batch on every rank: uint8 [n_channels, sizeof(p25fe_result_t)]
        (blocks are padded to the largest one for the all-gather)."""
        q = (self.n_channels + self.world - 1) // self.world
        item = result_local.shape[1]
        pad = torch.zeros((q, item), dtype=torch.uint8, device=result_local.device)
        pad[:self.n_local] = result_local
        if self.world == 1:
            return pad[:self.n_channels]
        out = torch.empty((self.world, q, item), dtype=torch.uint8, device=result_local.device)
        self.comm.all_gather(out, pad)
        rows = [out[r, :self.block(r, self.world, self.n_channels)[1] - self.block(r, self.world, self.n_channels)[0]]
                for r in range(self.world)]
        return torch.cat(rows, dim=0)
