"""ctypes binding of libp25fe.so (the C ABI of include/p25fe.h).

The library is the product: if it is missing or no gfx950 device is present every call
fails loudly -- there is no Python/NumPy compute fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libp25fe.so")
MAX_TAPS = 64
ABI_VERSION = 6
FMT_CF32, FMT_U8 = 0, 1

OK, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_FORMAT, ERR_NOMEM, ERR_JIT, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8
CLOCK_FIXED, CLOCK_TRACKING, CLOCK_TRACKING_RESLICE, CLOCK_CAUSAL_OK = 0, 1, 2, 0x100
SPECIALIZE_AUTO, SPECIALIZE_OFF, SPECIALIZE_REQUIRE, SPECIALIZE_FORCE = 0, -1, 1, 2
VARIANT_BUILTIN, VARIANT_SPECIALIZED, VARIANT_GENERIC = 0, 1, 2


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("n_channels", C.c_int32),
                ("n_decim_taps", C.c_int32), ("n_chan_taps", C.c_int32),
                ("decim_taps", C.c_float * MAX_TAPS), ("chan_taps", C.c_float * MAX_TAPS),
                ("symbol_clock", C.c_int32), ("specialize", C.c_int32),
                ("fm_deviation_hz", C.c_uint32), ("fm_sample_rate_hz", C.c_uint32), ("fm_gain", C.c_float),
                ("u8_scale", C.c_float), ("u8_offset", C.c_float), ("u8_lut_valid", C.c_int32), ("u8_lut", C.c_float * 256),
                ("decim_phase", C.c_int32), ("n_avg_taps", C.c_int32), ("avg_taps", C.c_float * MAX_TAPS)]


class WindowsStats(C.Structure):
    _fields_ = [("n_windows", C.c_uint64), ("ms_total", C.c_double), ("ms_h2d", C.c_double), ("ms_compute", C.c_double),
                ("pinned_input", C.c_int32), ("reserved", C.c_int32)]


class Anchor(C.Structure):
    _fields_ = [("s", C.c_int64), ("hi", C.c_float), ("mid", C.c_float), ("lo", C.c_float), ("valid", C.c_int32),
                ("period_d", C.c_int32), ("period_n", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n_baseband", C.c_uint64), ("n_dibits", C.c_uint64), ("n_sync", C.c_uint64),
                ("anchor_out", Anchor), ("first_event", C.c_int64), ("n_dibits_after_first", C.c_uint64),
                ("carry_end", C.c_int64), ("first_seg_end", C.c_int64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


ANCHOR_DTYPE = np.dtype([("s", "<i8"), ("hi", "<f4"), ("mid", "<f4"), ("lo", "<f4"), ("valid", "<i4"),
                         ("period_d", "<i4"), ("period_n", "<i4")])
RESULT_DTYPE = np.dtype([("n_baseband", "<u8"), ("n_dibits", "<u8"), ("n_sync", "<u8"), ("anchor_out", ANCHOR_DTYPE),
                         ("first_event", "<i8"), ("n_dibits_after_first", "<u8"), ("carry_end", "<i8"),
                         ("first_seg_end", "<i8"), ("flags", "<u4"), ("reserved", "<u4")])
NID_DTYPE = np.dtype([("raw", "<u8"), ("sync_pos", "<i8"), ("nac", "<u2"), ("duid", "u1"), ("n_errors", "u1"),
                      ("valid", "<i4")])
assert NID_DTYPE.itemsize == 24
CODE_STATS_DTYPE = np.dtype([("words", "<u8"), ("errs", "<u8"), ("fixed", "<u8"), ("size", "<u4"), ("reserved", "<u4")])
CHAN_STATS_DTYPE = np.dtype([("sig_power_dbm", "<f4"), ("locked", "<i4"), ("n_dibits", "<u8"), ("n_sync", "<u8"),
                             ("last_sync_pos", "<i8"), ("bch", CODE_STATS_DTYPE)])
assert CHAN_STATS_DTYPE.itemsize == 64
assert ANCHOR_DTYPE.itemsize == C.sizeof(Anchor) and RESULT_DTYPE.itemsize == C.sizeof(Result)

# every symbol include/p25fe.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "p25fe_default_config", "p25fe_create", "p25fe_destroy", "p25fe_strerror", "p25fe_last_hip_error", "p25fe_device",
    "p25fe_demod_u8", "p25fe_demod_cf32", "p25fe_slice", "p25fe_run_u8", "p25fe_run_cf32", "p25fe_resync",
    "p25fe_reset", "p25fe_state_size", "p25fe_state_export", "p25fe_state_import", "p25fe_demod_dev",
    "p25fe_slice_dev", "p25fe_run_dev", "p25fe_run_dev_pipelined", "p25fe_join_dev", "p25fe_shard_halo", "p25fe_shard_pass1", "p25fe_shard_pass2",
    "p25fe_shard_resolve", "p25fe_n_baseband", "p25fe_profile_enable", "p25fe_profile_read",
    "p25fe_predecim_dev", "p25fe_n_predecim", "p25fe_shard_resolve_dev", "p25fe_nid_dev",
    "p25fe_nid_batch_dev", "p25fe_chan_stats_dev", "p25fe_channelise_dev", "p25fe_nid",
    "p25fe_shard_pass1_main", "p25fe_shard_pass1_finish", "p25fe_shard_compact_dev", "p25fe_resync_at_dev",
    "p25fe_kernel_variant", "p25fe_specialize", "p25fe_specialize_log", "p25fe_run_host_windows",
    "p25fe_shard_pass1_head", "p25fe_shard_pipe_begin", "p25fe_shard_pipe_end", "p25fe_rx_stream", "p25fe_shard_head_check", "p25fe_shard_pass1_k1", "p25fe_streams_share_queue", "p25fe_shard_pass2_dev", "p25fe_shard_compact_from_dev", "p25fe_probe_variant", "p25fe_n_baseband_h",
]


class P25feError(RuntimeError):
    def __init__(self, status, msg, hip=0):
        super().__init__("p25fe: %s (status %d%s)" % (msg, status, ", hipError %d" % hip if hip else ""))
        self.status = status
        self.hip = hip


_LIB = None


def load():
    """Load libp25fe.so.  Raises if it has not been built -- never substitutes another path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libp25fe.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C p25rx_amd/csrc` (there is no CPU fallback)")
    # torch (plumbing: device memory, streams, RCCL) bundles its own libamdhip64.so.7.  Import it FIRST so
    # that libp25fe.so's NEEDED libamdhip64.so.7 binds to the runtime torch uses: two HIP runtimes in one
    # process do not share devices or pointers.  A host without torch binds to /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(os.environ.get("P25FE_LIB", LIB_PATH))      # P25FE_LIB: measurement builds (tools/ablate.sh) only
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    psz = C.POINTER(sz)
    L.p25fe_default_config.argtypes = [C.POINTER(Config)]
    L.p25fe_default_config.restype = None
    L.p25fe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.p25fe_destroy.argtypes = [vp]
    L.p25fe_destroy.restype = None
    L.p25fe_strerror.argtypes = [C.c_int]
    L.p25fe_strerror.restype = C.c_char_p
    L.p25fe_last_hip_error.argtypes = [vp]
    L.p25fe_device.argtypes = [vp]
    L.p25fe_kernel_variant.argtypes = [vp]
    L.p25fe_specialize.argtypes = [C.POINTER(Config), C.c_char_p, C.c_char_p, sz]
    L.p25fe_probe_variant.argtypes = [C.POINTER(Config)]
    L.p25fe_specialize_log.argtypes = [C.c_char_p, sz]
    L.p25fe_specialize_log.restype = sz
    L.p25fe_demod_u8.argtypes = [vp, vp, sz, vp, sz, psz, vp]
    L.p25fe_demod_cf32.argtypes = [vp, vp, sz, vp, sz, psz, vp]
    L.p25fe_slice.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, sz, vp]
    L.p25fe_run_u8.argtypes = [vp, vp, sz, vp, sz, vp]
    L.p25fe_run_cf32.argtypes = [vp, vp, sz, vp, sz, vp]
    L.p25fe_run_host_windows.argtypes = [vp, vp, C.c_int, sz, sz, vp, sz, vp, C.POINTER(WindowsStats)]
    L.p25fe_resync.argtypes = [vp]
    L.p25fe_reset.argtypes = [vp]
    L.p25fe_state_size.argtypes = [vp, psz]
    L.p25fe_state_export.argtypes = [vp, vp, sz, psz]
    L.p25fe_state_import.argtypes = [vp, vp, sz]
    L.p25fe_demod_dev.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp, sz, vp, vp]
    L.p25fe_slice_dev.argtypes = [vp, vp, sz, sz, sz, u64, vp, vp, sz, vp, vp, sz, vp, vp]
    L.p25fe_run_dev.argtypes = [vp, vp, C.c_int, sz, sz, vp, sz, vp, vp]
    L.p25fe_run_dev_pipelined.argtypes = [vp, vp, C.c_int, sz, sz, vp, sz, vp, vp]
    L.p25fe_join_dev.argtypes = [vp, vp]
    L.p25fe_shard_halo.argtypes = []
    L.p25fe_shard_halo.restype = sz
    L.p25fe_shard_pass1.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp, vp]
    L.p25fe_shard_pass1_main.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp]
    L.p25fe_shard_pass1_finish.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp, vp]
    L.p25fe_shard_compact_dev.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    L.p25fe_shard_pass1_head.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp]
    L.p25fe_shard_pipe_begin.argtypes = [vp, vp, C.POINTER(C.c_void_p)]
    L.p25fe_shard_head_check.argtypes = [C.c_void_p]
    L.p25fe_rx_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.p25fe_shard_pipe_end.argtypes = [vp, vp]
    L.p25fe_shard_pass1_k1.argtypes = [vp, vp, C.c_int, sz, sz, sz, u64, vp]
    L.p25fe_streams_share_queue.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
    L.p25fe_shard_pass2_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp, vp, sz, vp, vp, vp]
    L.p25fe_shard_compact_from_dev.argtypes = [vp, vp, sz, vp, sz, sz, vp, sz, vp]
    L.p25fe_shard_pass2.argtypes = [vp, vp, vp, sz, vp, vp]
    L.p25fe_shard_resolve.argtypes = [vp, vp, vp, sz, C.c_int, vp, vp]
    L.p25fe_resync_at_dev.argtypes = [vp, vp, sz, sz]
    L.p25fe_shard_resolve_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    L.p25fe_nid_dev.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp]
    L.p25fe_nid.argtypes = [vp, vp, sz, vp, vp, sz, vp]
    L.p25fe_nid_batch_dev.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, vp]
    L.p25fe_chan_stats_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.p25fe_channelise_dev.argtypes = [vp, vp, sz, sz, u64, vp, sz, vp]
    L.p25fe_profile_enable.argtypes = [vp, C.c_int]
    L.p25fe_profile_read.argtypes = [vp, C.POINTER(C.c_double * 4), C.POINTER(u64)]
    L.p25fe_predecim_dev.argtypes = [vp, vp, sz, sz, sz, u64, vp, sz, vp]
    L.p25fe_n_predecim.argtypes = [u64, sz]
    L.p25fe_n_predecim.restype = sz
    L.p25fe_n_baseband.argtypes = [u64, sz]
    L.p25fe_n_baseband.restype = sz
    L.p25fe_n_baseband_h.argtypes = [vp, u64, sz]
    L.p25fe_n_baseband_h.restype = sz
    _LIB = L
    return L


def default_config():
    cfg = Config()
    load().p25fe_default_config(C.byref(cfg))
    return cfg


def specialize_log():
    buf = C.create_string_buffer(1 << 16)
    load().p25fe_specialize_log(buf, len(buf))
    return buf.value.decode(errors="replace")


def make_config(n_channels=1, device=0, decim_taps=None, chan_taps=None, symbol_clock=0, specialize=SPECIALIZE_AUTO,
                fm_deviation_hz=None, fm_sample_rate_hz=None, fm_gain=None, u8_scale=None, u8_offset=None, u8_lut=None,
                decim_phase=None, avg_taps=None):
    """p25fe_config_t from keyword arguments (None = the build's default)."""
    cfg = default_config()
    cfg.device, cfg.n_channels, cfg.symbol_clock, cfg.specialize = device, n_channels, symbol_clock, specialize
    if decim_taps is not None:
        cfg.n_decim_taps = len(decim_taps)                       # the library rejects counts above P25FE_MAX_TAPS
        for i, v in enumerate(decim_taps[:MAX_TAPS]):
            cfg.decim_taps[i] = v
    if chan_taps is not None:
        cfg.n_chan_taps = len(chan_taps)
        for i, v in enumerate(chan_taps[:MAX_TAPS]):
            cfg.chan_taps[i] = v
    if fm_deviation_hz is not None:
        cfg.fm_deviation_hz = int(fm_deviation_hz)
    if fm_sample_rate_hz is not None:
        cfg.fm_sample_rate_hz = int(fm_sample_rate_hz)
    if fm_gain is not None:
        cfg.fm_gain = float(fm_gain)
    if u8_scale is not None:
        cfg.u8_scale = float(u8_scale)
    if u8_offset is not None:
        cfg.u8_offset = float(u8_offset)
    if u8_lut is not None:
        lut = np.asarray(u8_lut, dtype=np.float32)
        assert lut.shape == (256,)
        cfg.u8_lut_valid = 1
        for i in range(256):
            cfg.u8_lut[i] = lut[i]
    if decim_phase is not None:
        cfg.decim_phase = int(decim_phase)
    if avg_taps is not None:
        cfg.n_avg_taps = len(avg_taps)                           # the library rejects counts above P25FE_MAX_TAPS
        for i, v in enumerate(list(avg_taps)[:MAX_TAPS]):
            cfg.avg_taps[i] = v
    return cfg


def specialize(cfg, directory=None):
    """p25fe_specialize: compile (no GPU needed) and store the kernels for cfg's numbers; returns the file name ('' for the
    build's own numbers)."""
    L = load()
    out = C.create_string_buffer(4096)
    rc = L.p25fe_specialize(C.byref(cfg), directory.encode() if directory else None, out, len(out))
    if rc != OK:
        raise P25feError(rc, L.p25fe_strerror(rc).decode() + ": " + specialize_log()[-2000:])
    return out.value.decode()


def probe_variant(cfg):
    """p25fe_probe_variant: the kernels a handle made from cfg would run on this host (VARIANT_*); no GPU needed"""
    L = load()
    rc = L.p25fe_probe_variant(C.byref(cfg))
    if rc < 0:
        raise P25feError(rc, L.p25fe_strerror(rc).decode() + ": " + specialize_log()[-2000:])
    return rc


def check(L, h, rc):
    if rc != OK:
        hip = L.p25fe_last_hip_error(h) if (h and rc == ERR_HIP) else 0
        raise P25feError(rc, L.p25fe_strerror(rc).decode(), hip)
