"""ctypes binding of the CPU oracle (oracle/p25fe_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under p25rx_amd/ imports this module.
PARITY UNPINNED -- see the header of p25fe_oracle.c.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SPEC_JSON = os.path.join(ROOT, "tests", "golden", "spec.json")
MAX_TAPS = 64
NCOEF = 8


class Config(C.Structure):
    _fields_ = [
        ("decim", C.c_int32), ("t1", C.c_int32), ("t2", C.c_int32), ("boxcar", C.c_int32),
        ("sps", C.c_int32), ("peak_w", C.c_int32), ("sync_sign_mask", C.c_uint32),
        ("decim_taps", C.c_float * MAX_TAPS), ("chan_taps", C.c_float * MAX_TAPS),
        ("atan_c", C.c_float * NCOEF),
        ("fm_gain", C.c_float), ("u8_scale", C.c_float), ("boxcar_scale", C.c_float),
        ("pi", C.c_float), ("half_pi", C.c_float),
        ("inv_npos", C.c_float), ("inv_nneg", C.c_float), ("rho2_n", C.c_float),
        ("e_min", C.c_float), ("slice_frac", C.c_float),
        ("symbol_clock", C.c_int32), ("clk_lookahead", C.c_int32), ("clk_tol_shift", C.c_int32), ("clk_dmax_log2", C.c_int32),
        ("clk_interp", C.c_float * 256),
        ("u8_offset", C.c_float), ("u8_lut_valid", C.c_int32), ("u8_lut", C.c_float * 256),
        ("decim_phase", C.c_int32), ("n_avg", C.c_int32), ("avg_uniform", C.c_int32), ("avg_taps", C.c_float * MAX_TAPS),
    ]


def load_spec(path=SPEC_JSON):
    with open(path) as f:
        return json.load(f)


def fm_gain_from(deviation_hz, sample_rate_hz):
    """FmDemod::new(deviation, sample_rate), src/demod.rs:54 -> output scale (docs/SPEC.md 3.4): evaluated in double, rounded once"""
    return float(np.float32(float(sample_rate_hz) / (2.0 * np.pi * float(deviation_hz))))


def make_config(spec=None, decim_taps=None, chan_taps=None, symbol_clock=0, fm_deviation_hz=None, fm_sample_rate_hz=None,
                fm_gain=None, u8_scale=None, u8_offset=None, u8_lut=None, decim_phase=None, avg_taps=None):
    """The oracle's numbers: the build's (tests/golden/spec.json) unless overridden -- same keyword names and meaning as
    the p25fe_config_t fields (p25rx_amd/_lib.make_config)."""
    s = spec or load_spec()
    c = Config()
    c.symbol_clock = symbol_clock
    c.clk_lookahead, c.clk_tol_shift, c.clk_dmax_log2 = s["clk_lookahead"], s["clk_tol_shift"], s["clk_dmax_log2"]
    for q, row in enumerate(s["clk_interp"]):
        for k, v in enumerate(row):
            c.clk_interp[4 * q + k] = v
    c.decim, c.sps, c.boxcar, c.peak_w = s["decim"], s["sps"], s["boxcar_len"], s["sync_peak_w"]
    dt = list(s["decim_taps"] if decim_taps is None else decim_taps)
    ct = list(s["chan_taps"] if chan_taps is None else chan_taps)
    # docs/SPEC.md 3.3: a configured table IS the table of the evaluation length -- the build's 31 / 41, or 64 / 64 as soon as
    # either is longer -- with zero coefficients at the old end (the same filter; a non-finite sample then reaches that many taps)
    long_ = len(dt) > len(s["decim_taps"]) or len(ct) > len(s["chan_taps"])
    dt += [0.0] * ((64 if long_ else len(s["decim_taps"])) - len(dt))
    ct += [0.0] * ((64 if long_ else len(s["chan_taps"])) - len(ct))
    c.t1, c.t2 = len(dt), len(ct)
    for i, v in enumerate(dt):
        c.decim_taps[i] = v
    for i, v in enumerate(ct):
        c.chan_taps[i] = v
    for i, v in enumerate(s["atan_coeffs"]):
        c.atan_c[i] = v
    c.sync_sign_mask = s["sync_sign_mask"]
    c.fm_gain, c.u8_scale, c.boxcar_scale = s["fm_gain"], s["u8_scale"], s["boxcar_scale"]
    c.u8_offset = s["u8_offset"]
    if fm_gain is not None and fm_gain != 0.0:
        c.fm_gain = fm_gain
    elif fm_deviation_hz is not None or fm_sample_rate_hz is not None:
        c.fm_gain = fm_gain_from(fm_deviation_hz or s["fm_deviation_hz"], fm_sample_rate_hz or s["fm_sample_rate_hz"])
    if u8_scale is not None:
        c.u8_scale = u8_scale
    if u8_offset is not None:
        c.u8_offset = u8_offset
    if u8_lut is not None:
        c.u8_lut_valid = 1
        for i, v in enumerate(np.asarray(u8_lut, dtype=np.float32)):
            c.u8_lut[i] = v
    # ABI 5: Decimator::new(5)'s phase and MovingAverage::new(10) as a table (docs/SPEC.md 3.2, 3.5)
    c.decim_phase = s.get("decim_phase", 4) if decim_phase is None else int(decim_phase)
    at = np.asarray(s.get("avg_taps", [s["boxcar_scale"]] * s["boxcar_len"]) if avg_taps is None else avg_taps, dtype=np.float32)
    c.n_avg = len(at)
    c.avg_uniform = 1 if bool(np.all(at.view(np.uint32) == at.view(np.uint32)[0])) else 0
    for i, v in enumerate(at):
        c.avg_taps[i] = v
    c.pi, c.half_pi = s["pi"], s["half_pi"]
    c.inv_npos, c.inv_nneg = s["sync_inv_npos"], s["sync_inv_nneg"]
    c.rho2_n, c.e_min, c.slice_frac = s["sync_rho2_n"], s["sync_e_min"], s["slice_frac"]
    return c


def build(force=False):
    so = os.path.join(HERE, "libp25fe_oracle.so")
    src = os.path.join(HERE, "p25fe_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return so


def _bind(lib):
    vp, sz = C.c_void_p, C.c_size_t
    fp = C.POINTER(C.c_float)
    lib.p25o_demod_create.restype = vp
    lib.p25o_demod_create.argtypes = [C.POINTER(Config)]
    lib.p25o_demod_destroy.argtypes = [vp]
    lib.p25o_demod_u8.restype = sz
    lib.p25o_demod_u8.argtypes = [vp, vp, sz, vp, fp]
    lib.p25o_demod_cf32.restype = sz
    lib.p25o_demod_cf32.argtypes = [vp, vp, sz, vp, fp]
    lib.p25o_demod_cf32_stages.restype = sz
    lib.p25o_demod_cf32_stages.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.p25o_power_dbm.restype = C.c_float
    lib.p25o_power_dbm.argtypes = [vp, sz, C.c_float]
    lib.p25o_atan2f.restype = C.c_float
    lib.p25o_atan2f.argtypes = [C.POINTER(Config), C.c_float, C.c_float]
    lib.p25o_recv_create.restype = vp
    lib.p25o_recv_create.argtypes = [C.POINTER(Config)]
    lib.p25o_recv_destroy.argtypes = [vp]
    lib.p25o_recv_resync.argtypes = [vp]
    lib.p25o_recv_feed.restype = C.c_int
    lib.p25o_recv_feed.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), vp, vp, sz, C.POINTER(sz)]
    lib.p25o_recv_state.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_float * 3), C.POINTER(C.c_uint64)]
    lib.p25o_recv_record.argtypes = [vp, vp, sz]
    lib.p25o_recv_override.argtypes = [vp, vp, sz]
    lib.p25o_recv_n_det.restype = C.c_uint64
    lib.p25o_recv_n_det.argtypes = [vp]
    lib.p25o_reslice_table.argtypes = [vp, sz, C.c_int, vp]
    lib.p25o_predecim_create.restype = vp
    lib.p25o_predecim_create.argtypes = [vp, C.c_int, C.c_int]
    lib.p25o_predecim_destroy.argtypes = [vp]
    lib.p25o_predecim_feed.restype = sz
    lib.p25o_predecim_feed.argtypes = [vp, vp, sz, vp]
    lib.p25o_channelise.restype = sz
    lib.p25o_channelise.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, sz, sz, C.c_uint64, vp, sz]
    lib.p25o_nid_decode.restype = sz
    lib.p25o_nid_decode.argtypes = [vp, C.c_int, vp, sz, vp, vp, sz, vp]
    lib.p25o_run_cf32.restype = C.c_int64
    lib.p25o_run_cf32.argtypes = [C.POINTER(Config), vp, sz, vp, sz]
    lib.p25o_run_cf32_mt.restype = C.c_int64
    lib.p25o_run_cf32_mt.argtypes = [C.POINTER(Config), vp, sz, C.c_int]
    lib.p25o_has_fma.restype = C.c_int
    return lib


_LIB = None


def lib(path=None):
    global _LIB
    if path is not None:
        return _bind(C.CDLL(path))
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
        if not _LIB.p25o_has_fma():
            raise RuntimeError("oracle was built with -mfma but this CPU has no FMA")
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Demod:
    """demod::DemodTask state + one run-loop body per call (src/demod.rs:62-119)."""

    def __init__(self, cfg=None, libpath=None):
        self.L = lib(libpath)
        self.cfg = cfg or make_config()
        self.h = self.L.p25o_demod_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("p25o_demod_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.p25o_demod_destroy(self.h)
            self.h = None

    def feed_u8(self, data, want_power=False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        bb = np.empty(data.size // 2 // self.cfg.decim + 2, dtype=np.float32)
        p = C.c_float(0)
        n = self.L.p25o_demod_u8(self.h, _ptr(data), data.size, _ptr(bb), C.byref(p) if want_power else None)
        return (bb[:n].copy(), p.value) if want_power else bb[:n].copy()

    def feed_cf32(self, iq, want_power=False):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        bb = np.empty(iq.size // self.cfg.decim + 2, dtype=np.float32)
        p = C.c_float(0)
        n = self.L.p25o_demod_cf32(self.h, _ptr(iq), iq.size, _ptr(bb), C.byref(p) if want_power else None)
        return (bb[:n].copy(), p.value) if want_power else bb[:n].copy()

    def feed_cf32_stages(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        cap = iq.size // self.cfg.decim + 2
        ch = np.empty(cap, dtype=np.complex64)
        fm = np.empty(cap, dtype=np.float32)
        bb = np.empty(cap, dtype=np.float32)
        n = self.L.p25o_demod_cf32_stages(self.h, _ptr(iq), iq.size, _ptr(ch), _ptr(fm), _ptr(bb))
        return ch[:n].copy(), fm[:n].copy(), bb[:n].copy()


class PreDecim:
    """Stage 0 of config 3: 2.4 Msps -> 240 ksps, 10:1 decimating FIR (no reference counterpart)."""

    def __init__(self, spec=None, libpath=None):
        self.L = lib(libpath)
        s = spec or load_spec()
        taps = np.array(s["pre_taps"], dtype=np.float32)
        self.decim = s["pre_decim"]
        self.h = self.L.p25o_predecim_create(_ptr(taps), len(taps), self.decim)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.p25o_predecim_destroy(self.h)
            self.h = None

    def feed(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(iq.size // self.decim + 2, dtype=np.complex64)
        n = self.L.p25o_predecim_feed(self.h, _ptr(iq), iq.size, _ptr(out))
        return out[:n].copy()


def channelise(iq, n_hist=0, abs0=0, spec=None, libpath=None):
    """SPEC 3.11: wideband cf32 @ 2.4 Msps -> [192, n_out] complex64 @ 240 ksps (plain mix / filter / decimate in
    double precision).  iq holds n_hist samples of history followed by the owned range."""
    L = lib(libpath)
    s = spec or load_spec()
    taps = np.array(s["pre_taps"], dtype=np.float32)
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    n = iq.size - n_hist
    M, D = s["chz_channels"], s["pre_decim"]
    cap = n // D + 2
    out = np.zeros((M, cap), dtype=np.complex64)
    base = iq[n_hist:] if n_hist else iq
    k = L.p25o_channelise(_ptr(taps), len(taps), D, M, C.c_void_p(iq.ctypes.data + 8 * n_hist), n_hist, n, abs0,
                          _ptr(out), cap)
    return out[:, :k].copy()


class Recv:
    """Front half of MessageReceiver::feed + resync (src/recv.rs:136, 179, 207)."""

    def __init__(self, cfg=None, libpath=None):
        self.L = lib(libpath)
        self.cfg = cfg or make_config()
        self.h = self.L.p25o_recv_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.p25o_recv_destroy(self.h)
            self.h = None

    def resync(self):
        self.L.p25o_recv_resync(self.h)

    def feed(self, bb):
        bb = np.ascontiguousarray(bb, dtype=np.float32)
        # (a receiver that re-anchors on every sync word can emit more than n / sps dibits: up to one per peak_w + 1 samples under the
        # densest detections SPEC 3.7 allows, plus what an earlier feed left undecided)
        cap = bb.size // (self.cfg.peak_w + 1) + 8
        dib = np.empty(cap, dtype=np.uint8)
        scap = bb.size // (self.cfg.peak_w + 1) + 2
        spos = np.empty(scap, dtype=np.int64)
        sdib = np.empty(scap, dtype=np.uint64)
        nd, ns = C.c_size_t(0), C.c_size_t(0)
        rc = self.L.p25o_recv_feed(self.h, _ptr(bb), bb.size, _ptr(dib), cap, C.byref(nd),
                                   _ptr(spos), _ptr(sdib), scap, C.byref(ns))
        if rc:
            raise RuntimeError("oracle recv overflow")
        return dib[:nd.value].copy(), spos[:ns.value].copy(), sdib[:ns.value].copy()

    def state(self):
        t, v, s, nd = C.c_int64(), C.c_int(), C.c_int64(), C.c_uint64()
        thr = (C.c_float * 3)()
        self.L.p25o_recv_state(self.h, C.byref(t), C.byref(v), C.byref(s), C.byref(thr), C.byref(nd))
        return dict(t=t.value, valid=bool(v.value), s=s.value, hi=thr[0], mid=thr[1], lo=thr[2], n_dibits=nd.value)


CLK_DTYPE = np.dtype([("d", "<i8"), ("n", "<i8"), ("usable", "<i4"), ("f", "<i4")])


def recv_range(bb, cfg=None, resync_at=(), libpath=None):
    """One RESIDENT range through the receiver: bb from a fresh stream, lock dropped before every sample index of `resync_at`
    (ascending; p25fe_resync_at_dev).  symbol_clock 0 / 1: one pass of Recv.  symbol_clock 2 (SPEC 3.8c): pass 1 records every
    detection's backward clock, the table gives a detection without a usable one the clock of the interval that STARTS at it, pass
    2 slices with the table.  Returns (dibits, sync_pos, sync_dibit)."""
    cfg = cfg or make_config()
    bb = np.ascontiguousarray(bb, dtype=np.float32)
    cuts = [0] + [min(max(int(q), 0), bb.size) for q in sorted(resync_at)] + [bb.size]

    def one_pass(rec=None, ovr=None):
        r = Recv(cfg, libpath)
        keep = []
        if rec is not None:
            r.L.p25o_recv_record(r.h, _ptr(rec), rec.size)
        if ovr is not None:
            r.L.p25o_recv_override(r.h, _ptr(ovr), ovr.size)
            keep.append(ovr)
        outs = []
        for i in range(len(cuts) - 1):
            if i:
                r.resync()
            outs.append(r.feed(bb[cuts[i]:cuts[i + 1]]))
        nd = int(r.L.p25o_recv_n_det(r.h))
        base = np.cumsum([0] + [len(o[0]) for o in outs[:-1]])
        return (np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]),
                np.concatenate([o[2] for o in outs]) if outs else np.zeros(0, np.uint64), nd)

    if cfg.symbol_clock != 2:
        d, sp, sd, _ = one_pass()
        return d, sp, sd
    rec = np.zeros(bb.size // (cfg.peak_w + 1) + 2, dtype=CLK_DTYPE)
    _, _, _, nd = one_pass(rec=rec)
    tab = np.zeros(nd, dtype=CLK_DTYPE)
    lib(libpath).p25o_reslice_table(_ptr(rec), nd, cfg.sps, _ptr(tab))
    d, sp, sd, nd2 = one_pass(ovr=tab)
    assert nd2 == nd
    return d, sp, sd


NID_DTYPE = np.dtype([("raw", "<u8"), ("sync_pos", "<i8"), ("nac", "<u2"), ("duid", "u1"), ("n_errors", "u1"),
                      ("valid", "<i4")])


def nid_decode(dibits, sync_dibit, sync_pos=None, spec=None):
    """Network identifiers that follow the given sync events (SPEC 3.9); returns a NID_DTYPE array."""
    s = spec or load_spec()
    rows = np.array(s["nid_rows"], dtype=np.uint64)
    dibits = np.ascontiguousarray(dibits, dtype=np.uint8)
    sync_dibit = np.ascontiguousarray(sync_dibit, dtype=np.uint64)
    sp = None if sync_pos is None else np.ascontiguousarray(sync_pos, dtype=np.int64)
    out = np.zeros(len(sync_dibit), dtype=NID_DTYPE)
    lib().p25o_nid_decode(_ptr(rows), s["nid_t"], _ptr(dibits), dibits.size, _ptr(sync_dibit),
                          _ptr(sp) if sp is not None else None, len(sync_dibit), _ptr(out))
    return out


def power_dbm(samples, resistance=1.0):
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    return lib().p25o_power_dbm(_ptr(samples), samples.size, resistance)


def atan2f(y, x, cfg=None):
    cfg = cfg or make_config()
    return lib().p25o_atan2f(C.byref(cfg), y, x)


def run_cf32_mt(iq, nthreads, cfg=None, libpath=None):
    """Throughput helper for bench.py: `nthreads` independent time shards on as many threads; returns the dibit count."""
    L = lib(libpath)
    cfg = cfg or make_config()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    return int(L.p25o_run_cf32_mt(C.byref(cfg), _ptr(iq), iq.size, int(nthreads)))


def run_cf32(iq, cfg=None, libpath=None):
    """IQ -> dibits through the whole oracle path in the reference's 16384-sample chunks."""
    L = lib(libpath)
    cfg = cfg or make_config()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    cap = iq.size // (cfg.decim * cfg.sps) + 16
    dib = np.empty(cap, dtype=np.uint8)
    n = L.p25o_run_cf32(C.byref(cfg), _ptr(iq), iq.size, _ptr(dib), cap)
    if n < 0:
        raise RuntimeError("oracle run overflow")
    return dib[:n].copy()
