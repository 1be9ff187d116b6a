"""Handle-level Python view of the C ABI (include/p25fe.h).

`FrontEnd` owns one p25fe_t.  Host-buffer methods take/return NumPy arrays and mirror the
per-chunk bodies of DemodTask::run (src/demod.rs:70-117) and RecvTask::run
(src/recv.rs:148-150).  `*_dev` methods take torch CUDA tensors -- torch supplies device
memory and streams only; all arithmetic happens in libp25fe.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FMT_CF32, FMT_U8, RESULT_DTYPE, ANCHOR_DTYPE
CHZ_CHANNELS = 192                  # P25FE_CHZ_CHANNELS (include/p25fe_spec.h)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class FrontEnd:
    def __init__(self, n_channels=1, device=0, decim_taps=None, chan_taps=None, symbol_clock=0, **spec):
        """spec: the other p25fe_config_t fields by name (specialize, fm_deviation_hz, fm_sample_rate_hz, fm_gain, u8_scale,
        u8_offset, u8_lut, decim_phase, avg_taps) -- the run-time arguments of the reference's constructors
        (src/demod.rs:50, 52, 54, 83)."""
        self.L = _lib.load()
        # symbol_clock 0: fixed stride (the reference's receiver), 1: SPEC 3.8b, 2: 3.8b + 3.8c in run_dev / run_dev_pipelined / slice_dev
        cfg = _lib.make_config(n_channels=n_channels, device=device, decim_taps=decim_taps, chan_taps=chan_taps,
                               symbol_clock=symbol_clock, **spec)
        self.symbol_clock = symbol_clock & 0xff                  # (without P25FE_CLOCK_CAUSAL_OK)
        self.cfg = cfg
        self.C = n_channels
        self.device = device
        self.h = C.c_void_p()
        rc = self.L.p25fe_create(C.byref(cfg), C.byref(self.h))
        if rc == _lib.ERR_JIT:
            raise _lib.P25feError(rc, self.L.p25fe_strerror(rc).decode() + ": " + _lib.specialize_log()[-2000:])
        _lib.check(self.L, None, rc)

    @property
    def kernel_variant(self):
        """0: the library's own immediate-coefficient kernels, 1: kernels specialised for this handle's numbers, 2: generic"""
        return int(self.L.p25fe_kernel_variant(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.p25fe_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        _lib.check(self.L, self.h, rc)

    # ---- streaming, host buffers -------------------------------------------------------------
    def _demod(self, fn, arr, n_units, n_samples, want_power):
        cap = n_samples // 5 + 2
        bb = np.empty((self.C, cap), dtype=np.float32)
        n_out = C.c_size_t(0)
        pw = np.zeros(self.C, dtype=np.float32)
        self._chk(fn(self.h, _p(arr), n_units, _p(bb), cap, C.byref(n_out), _p(pw) if want_power else None))
        out = bb[:, :n_out.value]
        out = out[0].copy() if self.C == 1 else out.copy()
        if want_power:
            return out, (float(pw[0]) if self.C == 1 else pw)
        return out

    def demod_u8(self, data, want_power=False):
        """One DemodTask::run loop body on interleaved u8 I/Q; data shape [n_bytes] or [C, n_bytes]."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(self.C, -1)
        return self._demod(self.L.p25fe_demod_u8, data, data.shape[1], data.shape[1] // 2, want_power)

    def demod_cf32(self, iq, want_power=False):
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(self.C, -1)
        return self._demod(self.L.p25fe_demod_cf32, iq, iq.shape[1], iq.shape[1], want_power)

    def slice(self, bb, sync_cap=None):
        """RecvTask sample loop on baseband; returns (dibits, sync_pos, sync_dibit) (lists per channel if C > 1)."""
        bb = np.ascontiguousarray(bb, dtype=np.float32).reshape(self.C, -1)
        n = bb.shape[1]
        cap = n // 6 + 2                                         # hard ceiling: re-anchors are at least 6 samples apart (n // 10 + 1 is the undisturbed lock)
        scap = sync_cap if sync_cap is not None else n // 6 + 2
        dib = np.empty((self.C, cap), dtype=np.uint8)
        spos = np.empty((self.C, scap), dtype=np.int64)
        sdib = np.empty((self.C, scap), dtype=np.uint64)
        nd = (C.c_size_t * self.C)()
        ns = (C.c_size_t * self.C)()
        self._chk(self.L.p25fe_slice(self.h, _p(bb), n, _p(dib), cap, nd, _p(spos), _p(sdib), scap, ns))
        outs = [(dib[c, :nd[c]].copy(), spos[c, :min(ns[c], scap)].copy(), sdib[c, :min(ns[c], scap)].copy())
                for c in range(self.C)]
        return outs[0] if self.C == 1 else outs

    def _run(self, fn, arr, n_units, n_samples):
        cap = n_samples // 30 + 4                                # hard ceiling of the baseband (n // 5) under back-to-back re-anchors
        dib = np.empty((self.C, cap), dtype=np.uint8)
        nd = (C.c_size_t * self.C)()
        self._chk(fn(self.h, _p(arr), n_units, _p(dib), cap, nd))
        outs = [dib[c, :nd[c]].copy() for c in range(self.C)]
        return outs[0] if self.C == 1 else outs

    def run_u8(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(self.C, -1)
        return self._run(self.L.p25fe_run_u8, data, data.shape[1], data.shape[1] // 2)

    def run_cf32(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(self.C, -1)
        return self._run(self.L.p25fe_run_cf32, iq, iq.shape[1], iq.shape[1])

    def run_host_windows(self, iq, window=0, fmt=None):
        """p25fe_run_host_windows: a LONG host capture through the path as a pipeline of windows (H2D copy | kernels | dibits
        back).  iq: numpy array (pageable: staged by the library) or a CPU torch tensor (pinned: copied from directly) --
        complex64 / float32 pairs (cf32) or uint8 pairs (u8), [n] or [C, n].  Returns (dibits per channel, stats dict)."""
        if hasattr(iq, "data_ptr"):                                  # torch CPU tensor (possibly pinned)
            import torch
            assert not iq.is_cuda and iq.is_contiguous()
            is_u8 = iq.dtype == torch.uint8
            n_el = iq.numel() // self.C
            n = n_el // 2
            ptr = C.c_void_p(iq.data_ptr())
        else:
            is_u8 = iq.dtype == np.uint8
            iq = np.ascontiguousarray(iq if is_u8 else iq.view(np.float32) if iq.dtype == np.complex64 else iq.astype(np.float32))
            n = iq.size // self.C // 2
            ptr = _p(iq)
        fmt = (FMT_U8 if is_u8 else FMT_CF32) if fmt is None else fmt
        cap = n // 30 + 4
        dib = np.empty((self.C, cap), dtype=np.uint8)
        nd = (C.c_size_t * self.C)()
        st = _lib.WindowsStats()
        self._chk(self.L.p25fe_run_host_windows(self.h, ptr, fmt, n, int(window), _p(dib), cap, nd, C.byref(st)))
        outs = [dib[c, :nd[c]].copy() for c in range(self.C)]
        stats = dict(n_windows=int(st.n_windows), ms_total=st.ms_total, ms_h2d=st.ms_h2d, ms_compute=st.ms_compute,
                     pinned_input=bool(st.pinned_input))
        return (outs[0] if self.C == 1 else outs), stats

    def resync(self):
        self._chk(self.L.p25fe_resync(self.h))

    def reset(self):
        self._chk(self.L.p25fe_reset(self.h))

    def resync_at_dev(self, idx):
        """Lock drops for the NEXT receiver-running device call: idx = int64 device tensor [n] or [C, n] of ascending absolute
        baseband indices (lock is dropped before each); None / empty cancels.  The tensor is kept alive by the wrapper."""
        if idx is None or idx.numel() == 0:
            self._rs = None
            self._chk(self.L.p25fe_resync_at_dev(self.h, None, 0, 0))
            return
        import torch
        assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()
        if idx.dim() == 1:
            idx = idx.unsqueeze(0).expand(self.C, -1).contiguous() if self.C > 1 else idx.unsqueeze(0)
        assert idx.shape[0] == self.C
        self._rs = idx
        self._chk(self.L.p25fe_resync_at_dev(self.h, C.c_void_p(idx.data_ptr()), idx.shape[1], idx.stride(0)))

    def state_export(self):
        n = C.c_size_t(0)
        self._chk(self.L.p25fe_state_size(self.h, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        self._chk(self.L.p25fe_state_export(self.h, _p(buf), buf.size, C.byref(n)))
        return buf

    def state_import(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self.L.p25fe_state_import(self.h, _p(blob), blob.size))

    # ---- device-resident ranges (torch tensors as plumbing) -------------------------------------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _fmt_of(t):
        import torch
        if t.dtype == torch.float32:
            return FMT_CF32, t.shape[-2] if t.dim() == 3 else t.shape[0]
        if t.dtype == torch.uint8:
            return FMT_U8, t.shape[-2] if t.dim() == 3 else t.shape[0]
        raise TypeError("IQ tensor must be float32 [C, n, 2] or uint8 [C, n, 2]")

    def _iq_view(self, iq):
        """iq: [n, 2] or [C, n, 2] contiguous per channel; returns (fmt, n, ch_stride)."""
        import torch
        assert iq.is_cuda and iq.shape[-1] == 2
        if iq.dim() == 2:
            iq = iq.unsqueeze(0)
        assert iq.shape[0] == self.C and iq.stride(2) == 1 and iq.stride(1) == 2
        fmt = FMT_CF32 if iq.dtype == torch.float32 else FMT_U8
        if iq.dtype not in (torch.float32, torch.uint8):
            raise TypeError("IQ tensor must be float32 or uint8")
        return fmt, iq.shape[1], (iq.stride(0) // 2 if self.C > 1 else iq.shape[1])

    @staticmethod
    def dibit_cap(n_iq):
        """Row size for the dibits of n_iq input samples: n / 50 plus proportional slack (200 ppm) for the transmitter's
        symbol clock, which a receiver that re-anchors on every sync word follows.  The row's stride is its capacity: the
        kernels never store past it and p25fe_result_t.n_dibits stays exact."""
        return (n_iq // 50 + n_iq // 250000 + 64 + 15) // 16 * 16

    def run_dev(self, iq, dibits=None, result=None):
        """Fresh-stream IQ -> dibits on device.  Returns (dibits[C, cap] uint8 cuda, result uint8 tensor)."""
        import torch
        fmt, n, stride = self._iq_view(iq)
        cap = self.dibit_cap(n)
        if dibits is None:
            dibits = torch.empty((self.C, cap), dtype=torch.uint8, device=iq.device)
        if result is None:
            result = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=iq.device)
        self._chk(self.L.p25fe_run_dev(self.h, C.c_void_p(iq.data_ptr()), fmt, stride, n, C.c_void_p(dibits.data_ptr()),
                                       dibits.stride(0), C.c_void_p(result.data_ptr()), self._stream()))
        return dibits, result

    def run_dev_pipelined(self, iq, dibits=None, result=None):
        """run_dev for a sequence of captures: the receive kernels of this call overlap the next call's K1 (they run
        on a stream of the handle).  Outputs are complete after join_dev() + a synchronisation of the stream."""
        import torch
        fmt, n, stride = self._iq_view(iq)
        cap = self.dibit_cap(n)
        if dibits is None:
            dibits = torch.empty((self.C, cap), dtype=torch.uint8, device=iq.device)
        if result is None:
            result = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=iq.device)
        self._chk(self.L.p25fe_run_dev_pipelined(self.h, C.c_void_p(iq.data_ptr()), fmt, stride, n,
                                                 C.c_void_p(dibits.data_ptr()), dibits.stride(0),
                                                 C.c_void_p(result.data_ptr()), self._stream()))
        # the receive kernels write these on the handle's own stream, which torch's caching allocator does not see: keep
        # the outputs of the calls still in flight alive even if the caller drops them (released by join_dev)
        self._inflight = (getattr(self, "_inflight", ()) + ((dibits, result),))[-2:]
        return dibits, result

    def join_dev(self):
        """the current stream waits for every receive kernel run_dev_pipelined has enqueued"""
        self._chk(self.L.p25fe_join_dev(self.h, self._stream()))
        self._inflight = ()

    def demod_dev(self, iq, n_hist=0, abs0=0, bb=None, want_power=False, offset=0):
        """stages 1-5 on a device range; `offset` = index of the first owned sample inside `iq` (>= n_hist)."""
        import torch
        fmt, n_total, stride = self._iq_view(iq)
        n = n_total - offset
        nb = self.L.p25fe_n_baseband_h(self.h, abs0, n)
        if bb is None:
            bb = torch.empty((self.C, (nb + 7) // 4 * 4), dtype=torch.float32, device=iq.device)
        pw = torch.empty(self.C, dtype=torch.float32, device=iq.device) if want_power else None
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_demod_dev(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n, abs0,
                                         C.c_void_p(bb.data_ptr()), bb.stride(0),
                                         C.c_void_p(pw.data_ptr()) if want_power else None, self._stream()))
        return (bb, nb, pw) if want_power else (bb, nb)

    def predecim_dev(self, iq, n_hist=0, abs0=0, offset=0, out=None):
        """Stage 0 (config 3): cf32 @ 2.4 Msps [n, 2] or [C, n, 2] -> cf32 @ 240 ksps [C, n_out, 2] on device."""
        import torch
        fmt, n_total, stride = self._iq_view(iq)
        assert fmt == FMT_CF32
        n = n_total - offset
        no = self.L.p25fe_n_predecim(abs0, n)
        if out is None:
            out = torch.empty((self.C, (no + 3) // 2 * 2, 2), dtype=torch.float32, device=iq.device)
        self._chk(self.L.p25fe_predecim_dev(self.h, C.c_void_p(iq.data_ptr() + 8 * offset), stride, n_hist, n, abs0,
                                            C.c_void_p(out.data_ptr()), out.stride(0) // 2, self._stream()))
        return out, no

    def channelise_dev(self, iq, n_hist=0, abs0=0, offset=0, out=None):
        """SPEC 3.11: wideband cf32 @ 2.4 Msps [n, 2] (device) -> [192, n_out (padded), 2] cf32 @ 240 ksps."""
        import torch
        assert iq.dtype == torch.float32 and iq.dim() == 2 and iq.shape[1] == 2
        n = iq.shape[0] - offset
        no = self.L.p25fe_n_predecim(abs0, n)
        if out is None:
            out = torch.empty((CHZ_CHANNELS, max(64, (no + 63) // 64 * 64), 2), dtype=torch.float32, device=iq.device)
        self._chk(self.L.p25fe_channelise_dev(self.h, C.c_void_p(iq.data_ptr() + 8 * offset), n_hist, n, abs0,
                                              C.c_void_p(out.data_ptr()), out.stride(0) // 2, self._stream()))
        return out, no

    def slice_dev(self, bb, n_bb, n_hist_bb=0, abs_bb0=0, anchor_in=None, offset=0, sync_cap=0):
        import torch
        dev = bb.device
        if bb.dim() == 1:
            bb = bb.unsqueeze(0)
        cap = (n_bb // 10 + 64 + 15) // 16 * 16
        dib = torch.empty((self.C, cap), dtype=torch.uint8, device=dev)
        res = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        spos = torch.empty((self.C, max(sync_cap, 1)), dtype=torch.int64, device=dev)
        sdib = torch.empty((self.C, max(sync_cap, 1)), dtype=torch.int64, device=dev)
        a_in = None
        if anchor_in is not None:
            a_in = torch.from_numpy(np.frombuffer(np.asarray(anchor_in, dtype=ANCHOR_DTYPE).tobytes(), dtype=np.uint8).copy()).to(dev)
        self._chk(self.L.p25fe_slice_dev(self.h, C.c_void_p(bb.data_ptr() + 4 * offset), bb.stride(0), n_hist_bb, n_bb,
                                         abs_bb0, C.c_void_p(a_in.data_ptr()) if a_in is not None else None,
                                         C.c_void_p(dib.data_ptr()), dib.stride(0),
                                         C.c_void_p(spos.data_ptr()) if sync_cap else None,
                                         C.c_void_p(sdib.data_ptr()) if sync_cap else None, sync_cap,
                                         C.c_void_p(res.data_ptr()), self._stream()))
        return dib, res, spos, sdib

    def nid_dev(self, dibits, n_dibits, sync_dibit, sync_pos=None, n_sync=None):
        """Network identifiers after the sync events of ONE channel: dibits uint8 [>= n_dibits], sync_dibit int64/uint64
        [n_sync] (device tensors) -> uint8 tensor [n_sync, 24] (view with _lib.NID_DTYPE after .cpu())."""
        import torch
        n_sync = int(sync_dibit.numel() if n_sync is None else n_sync)
        out = torch.empty((max(n_sync, 1), 24), dtype=torch.uint8, device=dibits.device)
        self._chk(self.L.p25fe_nid_dev(self.h, C.c_void_p(dibits.data_ptr()), int(n_dibits), C.c_void_p(sync_dibit.data_ptr()),
                                       C.c_void_p(sync_pos.data_ptr()) if sync_pos is not None else None, n_sync,
                                       C.c_void_p(out.data_ptr()), self._stream()))
        return out[:n_sync]

    def nid_batch_dev(self, dibits, results, sync_dibit, sync_pos=None):
        """Channel batch: dibits [C, dibit_stride] uint8, results uint8 [C, sizeof(p25fe_result_t)], sync_dibit /
        sync_pos [C, sync_stride] (all device tensors, as written by slice_dev) -> uint8 tensor [C, sync_stride, 24]
        (rows k >= results[c].n_sync are left untouched: zero)."""
        import torch
        sync_stride = sync_dibit.shape[1]
        out = torch.zeros((self.C, max(sync_stride, 1), 24), dtype=torch.uint8, device=dibits.device)
        self._chk(self.L.p25fe_nid_batch_dev(self.h, C.c_void_p(dibits.data_ptr()), dibits.stride(0),
                                             C.c_void_p(results.data_ptr()), C.c_void_p(sync_dibit.data_ptr()),
                                             C.c_void_p(sync_pos.data_ptr()) if sync_pos is not None else None,
                                             sync_stride, C.c_void_p(out.data_ptr()), self._stream()))
        return out[:, :sync_stride]

    def chan_stats_dev(self, results, nid=None, power_dbm=None):
        """results [C, sizeof(p25fe_result_t)], nid [C, sync_stride, 24] (optional), power_dbm [C] float32
        (optional) -> uint8 tensor [C, 64] (view with _lib.CHAN_STATS_DTYPE after .cpu())."""
        import torch
        out = torch.empty((self.C, 64), dtype=torch.uint8, device=results.device)
        self._chk(self.L.p25fe_chan_stats_dev(self.h, C.c_void_p(results.data_ptr()),
                                              C.c_void_p(nid.data_ptr()) if nid is not None else None,
                                              nid.shape[1] if nid is not None else 0,
                                              C.c_void_p(power_dbm.data_ptr()) if power_dbm is not None else None,
                                              C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def profile_enable(self, on=True):
        """True / 1: events around every kernel; 2: around K1 only; 3: around K1 on every 8th call; False / 0: off."""
        self._chk(self.L.p25fe_profile_enable(self.h, int(on)))

    def profile_read(self):
        """-> (ms per kernel [K1 front end, K2 sync, K3 scan, K4 slice] summed over calls, n_calls)."""
        ms = (C.c_double * 4)()
        n = C.c_uint64(0)
        self._chk(self.L.p25fe_profile_read(self.h, C.byref(ms), C.byref(n)))
        return [ms[i] for i in range(4)], n.value

    def n_baseband(self, abs0, n):
        """baseband samples n input samples from absolute index abs0 give, with THIS handle's decimator phase"""
        return int(self.L.p25fe_n_baseband_h(self.h, abs0, n))

    def shard_halo(self):
        return self.L.p25fe_shard_halo()

    def shard_pass1(self, iq, offset, n_hist, abs0, result=None):
        import torch
        fmt, n_total, stride = self._iq_view(iq)
        n = n_total - offset
        if result is None:
            result = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=iq.device)
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_shard_pass1(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n, abs0,
                                           C.c_void_p(result.data_ptr()), self._stream()))
        return result

    def shard_pass1_main(self, iq, offset, n_hist, abs0):
        """K1 over everything that does not need the left halo (may run while the halo is still on the wire)."""
        fmt, n_total, stride = self._iq_view(iq)
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_shard_pass1_main(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n_total - offset, abs0,
                                                self._stream()))

    def shard_pass1_finish(self, iq, offset, n_hist, abs0, result=None):
        """The shard's head (needs the halo) + sync detection + scan; same arguments as shard_pass1_main."""
        import torch
        fmt, n_total, stride = self._iq_view(iq)
        if result is None:
            result = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=iq.device)
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_shard_pass1_finish(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n_total - offset, abs0,
                                                  C.c_void_p(result.data_ptr()), self._stream()))
        return result

    def shard_pass1_head(self, iq, offset, n_hist, abs0):
        """The shard's head segment alone (needs the halo); may run on another stream beside shard_pass1_main's launch."""
        fmt, n_total, stride = self._iq_view(iq)
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_shard_pass1_head(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n_total - offset, abs0,
                                                self._stream()))

    def shard_head_check(self):
        """after a synchronise: raises P25feError(ERR_TIMEOUT) if a detection gave up waiting for a head segment (p25fe_shard_head_check)"""
        self._chk(self.L.p25fe_shard_head_check(self.h))

    def shard_pipe_begin(self):
        """p25fe_shard_pipe_begin on torch's current stream -> the handle's receive stream as a torch stream: shard_pass1_main stays
        on the current stream, every later pass of the step is enqueued under `with torch.cuda.stream(rx)`; then shard_pipe_end()."""
        import torch
        rx = C.c_void_p()
        self._chk(self.L.p25fe_shard_pipe_begin(self.h, self._stream(), C.byref(rx)))
        return torch.cuda.ExternalStream(rx.value)

    def shard_pipe_end(self, last=None):
        """last: the torch stream the step's final pass was enqueued on (default: the receive stream)"""
        self._chk(self.L.p25fe_shard_pipe_end(self.h, C.c_void_p(last.cuda_stream) if last is not None else None))

    def shard_pass1_k1(self, iq, offset, n_hist, abs0):
        """The whole front end of pass 1 (main + head) in one launch; shard_pass1_finish then runs detection + scan only."""
        fmt, n_total, stride = self._iq_view(iq)
        ptr = iq.data_ptr() + offset * (8 if fmt == FMT_CF32 else 2)
        self._chk(self.L.p25fe_shard_pass1_k1(self.h, C.c_void_p(ptr), fmt, stride, n_hist, n_total - offset, abs0, self._stream()))

    def streams_share_queue(self, a, b):
        """do the torch streams a and b (None: the NULL stream) sit on one hardware queue?  (p25fe_streams_share_queue; synchronises both)"""
        sh = C.c_int(0)
        pa = C.c_void_p(a.cuda_stream) if a is not None else None
        pb = C.c_void_p(b.cuda_stream) if b is not None else None
        self._chk(self.L.p25fe_streams_share_queue(self.h, pa, pb, C.byref(sh)))
        return bool(sh.value)

    def shard_pass2_dev(self, summ_all, d_bb0, d_bbn, rank, n_bb, dibits=None, dup=None, result=None):
        """Pass 2 with the combine inside it (p25fe_shard_pass2_dev): summ_all uint8 [n_shards, sizeof(result)] on the device.
        Returns (dibits, result, anchors uint8 [n_shards, sizeof(anchor)], offsets int64 [n_shards + 1])."""
        import torch
        n = summ_all.shape[0]
        dev = summ_all.device
        cap = (n_bb // 10 + 64 + 15) // 16 * 16
        dib = dibits if dibits is not None else torch.empty((1, cap), dtype=torch.uint8, device=dev)
        if result is None:
            result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        anchors = torch.empty((n, ANCHOR_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
        self._chk(self.L.p25fe_shard_pass2_dev(self.h, C.c_void_p(summ_all.data_ptr()), C.c_void_p(d_bb0.data_ptr()),
                                               C.c_void_p(d_bbn.data_ptr()), n, rank, C.c_void_p(anchors.data_ptr()),
                                               C.c_void_p(offsets.data_ptr()), C.c_void_p(dib.data_ptr()), dib.stride(0),
                                               C.c_void_p(dup.data_ptr()) if dup is not None else None,
                                               C.c_void_p(result.data_ptr()), self._stream()))
        return dib, result, anchors, offsets

    def shard_compact_from_dev(self, gathered, offsets, first, out):
        """shard_compact_dev for shards first .. n - 1 only (rank 0 has sliced its own shard into `out` already)."""
        self._chk(self.L.p25fe_shard_compact_from_dev(self.h, C.c_void_p(gathered.data_ptr()), gathered.shape[1],
                                                      C.c_void_p(offsets.data_ptr()), first, gathered.shape[0],
                                                      C.c_void_p(out.data_ptr()), out.numel(), self._stream()))
        return out

    def shard_compact_dev(self, gathered, offsets, out):
        """gathered uint8 [n_shards, cap] (all-gathered shard streams), offsets int64 [n_shards + 1] -> out uint8 [total]."""
        self._chk(self.L.p25fe_shard_compact_dev(self.h, C.c_void_p(gathered.data_ptr()), gathered.shape[1],
                                                 C.c_void_p(offsets.data_ptr()), gathered.shape[0],
                                                 C.c_void_p(out.data_ptr()), out.numel(), self._stream()))
        return out

    def shard_resolve_dev(self, summ_all, d_bb0, d_bbn, anchors=None, offsets=None):
        """Device-side combine: summ_all uint8 [n_shards, sizeof(result)], d_bb0 / d_bbn int64 device tensors.
        Returns (anchors uint8 [n_shards, sizeof(anchor)], offsets int64 [n_shards + 1], last = total) without
        synchronising."""
        import torch
        n = summ_all.shape[0]
        if anchors is None:
            anchors = torch.empty((n, ANCHOR_DTYPE.itemsize), dtype=torch.uint8, device=summ_all.device)
        if offsets is None:
            offsets = torch.empty(n + 1, dtype=torch.int64, device=summ_all.device)
        self._chk(self.L.p25fe_shard_resolve_dev(self.h, C.c_void_p(summ_all.data_ptr()), C.c_void_p(d_bb0.data_ptr()),
                                                 C.c_void_p(d_bbn.data_ptr()), n, C.c_void_p(anchors.data_ptr()),
                                                 C.c_void_p(offsets.data_ptr()), self._stream()))
        return anchors, offsets

    def shard_pass2(self, anchor_in, n_bb, device, result=None, dibits=None):
        """anchor_in: NumPy structured anchor(s) (host) or a uint8 device tensor holding one p25fe_anchor_t per channel."""
        import torch
        cap = (n_bb // 10 + 64 + 15) // 16 * 16
        dib = dibits if dibits is not None else torch.empty((self.C, cap), dtype=torch.uint8, device=device)
        if result is None:
            result = torch.empty((self.C, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device)
        if isinstance(anchor_in, torch.Tensor):
            a_in = anchor_in
        else:
            a_in = torch.from_numpy(np.frombuffer(np.asarray(anchor_in, dtype=ANCHOR_DTYPE).tobytes(), dtype=np.uint8).copy()).to(device)
        self._chk(self.L.p25fe_shard_pass2(self.h, C.c_void_p(a_in.data_ptr()), C.c_void_p(dib.data_ptr()), dib.stride(0),
                                           C.c_void_p(result.data_ptr()), self._stream()))
        return dib, result

    def shard_resolve(self, summaries, bb0, bbn):
        """Host combine of per-shard summaries (np structured RESULT_DTYPE, time order) -> (anchor_in, dibit_offset)."""
        summaries = np.ascontiguousarray(summaries, dtype=RESULT_DTYPE)
        bb0 = np.ascontiguousarray(bb0, dtype=np.uint64)
        bbn = np.ascontiguousarray(bbn, dtype=np.uint64)
        n = len(summaries)
        anc = np.zeros(n, dtype=ANCHOR_DTYPE)
        off = np.zeros(n + 1, dtype=np.uint64)                   # [n] = total dibits of the capture
        self._chk(self.L.p25fe_shard_resolve(_p(summaries), _p(bb0), _p(bbn), n, self.symbol_clock, _p(anc), _p(off)))
        return anc, off


def parse_results(t):
    """uint8 result tensor [C, sizeof(p25fe_result_t)] -> NumPy structured array (syncs the stream)."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=RESULT_DTYPE).copy()


def n_baseband(abs0, n):
    return _lib.load().p25fe_n_baseband(abs0, n)
