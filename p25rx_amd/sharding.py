"""Time-sharded captures across the GPUs of one node (BASELINE.json config 5).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  Rank r owns the contiguous time range [r*n, (r+1)*n) of one long capture.  The path
has exactly two exchange steps, both tiny:

  1. filter-state overlap: the last `halo` IQ samples of rank r go to rank r+1 (point-to-point;
     on MI355X that is one xGMI link, ~12.5 KB),
  2. symbol-timing carry: every rank's 56-byte shard summary is all-gathered; each rank then
     resolves its own carry-in anchor and dibit offset with p25fe_shard_resolve (host logic).

The reduced dibit stream stays sharded in HBM; `gather_counts` gives every rank the global
dibit layout so that rank 0 (or any consumer) can fetch sum(counts) bytes.

The reference has no distributed code at all; the serial state it carries
across chunks (src/demod.rs:25-40, MessageReceiver's lock) is what steps 1 and 2 hand over.
"""
import numpy as np

from ._lib import RESULT_DTYPE


class TimeShard:
    """Per-rank driver.  `fe` is a FrontEnd (or a test double with the same shard_* methods)."""

    def __init__(self, fe, rank, world, n_per_rank, dist=None):
        self.fe, self.rank, self.world, self.n, self.dist = fe, rank, world, n_per_rank, dist
        self.halo = int(fe.shard_halo())
        self.abs0 = rank * n_per_rank
        from .frontend import n_baseband
        self.bb0 = [n_baseband(0, r * n_per_rank) for r in range(world)]
        self.bbn = [n_baseband(r * n_per_rank, n_per_rank) for r in range(world)]

    def alloc(self, torch, device, dtype):
        """[halo | owned] buffer; the halo part is filled by exchange_halo()."""
        return torch.zeros((self.halo + self.n, 2), dtype=dtype, device=device)

    def exchange_halo(self, buf):
        """Step 1: last `halo` samples of my range -> rank+1; rank-1's arrive in buf[:halo]."""
        if self.world == 1:
            return
        d = self.dist
        ops = []
        if self.rank + 1 < self.world:
            ops.append(d.P2POp(d.isend, buf[self.n:], self.rank + 1))
        if self.rank > 0:
            ops.append(d.P2POp(d.irecv, buf[:self.halo], self.rank - 1))
        for w in d.batch_isend_irecv(ops):
            w.wait()

    def pass1(self, buf, result):
        h = self.halo if self.rank > 0 else 0
        return self.fe.shard_pass1(buf[self.halo - h:], offset=h, n_hist=h, abs0=self.abs0, result=result)

    def exchange_summaries(self, result, summ_all):
        """Step 2: all-gather the per-rank summaries (one p25fe_result_t each) and resolve the carry."""
        if self.world > 1:
            self.dist.all_gather_into_tensor(summ_all.view(-1), result.view(-1))
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
        """One pass of the hot path over my shard; returns (dibit_offset_of_my_shard, summaries)."""
        self.exchange_halo(buf)
        self.pass1(buf, result)
        summ, anchors, offsets = self.exchange_summaries(result, summ_all)
        self.pass2(anchors, buf.device, result, dibits)
        return int(offsets[self.rank]), summ

    # ---- same step without any host synchronisation: the carry is resolved by a one-thread kernel ----------
    def setup_device(self, torch, device):
        self.d_bb0 = torch.tensor(self.bb0, dtype=torch.int64, device=device)
        self.d_bbn = torch.tensor(self.bbn, dtype=torch.int64, device=device)
        self.d_anchors = self.d_offsets = None

    def step_device(self, buf, result, summ_all, dibits):
        """exchange_halo -> pass 1 -> all_gather -> k_shard_resolve -> pass 2, all enqueued on the current stream.
        Returns the device tensor of per-shard dibit offsets (read it after the timed region)."""
        self.exchange_halo(buf)
        self.pass1(buf, result)
        if self.world > 1:
            self.dist.all_gather_into_tensor(summ_all.view(-1), result.view(-1))
        else:
            summ_all.copy_(result)
        self.d_anchors, self.d_offsets = self.fe.shard_resolve_dev(summ_all, self.d_bb0, self.d_bbn, self.d_anchors,
                                                                   self.d_offsets)
        self.fe.shard_pass2(self.d_anchors[self.rank:self.rank + 1], self.bbn[self.rank], buf.device, result=result,
                            dibits=dibits)
        return self.d_offsets
