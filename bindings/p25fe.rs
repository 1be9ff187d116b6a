//! bindings/p25fe.rs -- Rust side of the C ABI of `include/p25fe.h` / `include/p25fe_rccl.h` (ABI version 6).
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no Rust toolchain.  This is the file a
//! maintainer of kchmck/p25rx drops in as `src/p25fe.rs` (`mod p25fe;` in `src/main.rs:55-64`); the layouts below are the
//! `#[repr(C)]` twins of the C structs, and `p25rx_amd/_lib.py` holds the same layouts as ctypes / NumPy types whose sizes
//! the tests assert.  The tested host code of this repository is the C++ mirror (`p25rx_amd/host/p25fe_host.hpp`) and the
//! Python one (`p25rx_amd/demod.py`, `recv.py`), which have the shape of `DemodTask` / `RecvTask` below.
//!
//! What it replaces in the reference (file:line in kchmck/p25rx):
//!   * `DemodTask::new`  src/demod.rs:44-59   -> `FrontEnd::new` (tables, FM deviation / rate, the u8 table: all run-time)
//!   * `DemodTask::run`  src/demod.rs:62-119  -> `DemodTask::run` below: one `p25fe_demod_u8` per buffer
//!   * `RecvTask::run` sample loop  src/recv.rs:148-150, 204-210 -> `FrontEnd::slice` (or keep `MessageReceiver::feed`)
//!   * `MessageReceiver::resync`    src/recv.rs:136, 179         -> `FrontEnd::resync`
#![allow(dead_code, non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const ABI_VERSION: i32 = 6;
pub const MAX_TAPS: usize = 64;
pub const FMT_CF32: c_int = 0;
pub const FMT_U8: c_int = 1;
pub const CLOCK_FIXED: i32 = 0;
pub const CLOCK_TRACKING: i32 = 1;
/// CLOCK_TRACKING, and the calls that hold a whole range (run_dev, run_dev_pipelined, slice_dev) re-slice the first frame of a lock
/// run with the period the next sync word confirms (docs/SPEC.md 3.8c).  The calls that see the stream in pieces (slice, run_u8 / run_cf32,
/// run_host_windows, the time-shard passes) cannot: on such a handle they return ERR_ARG (ABI 6) unless CLOCK_CAUSAL_OK was or-ed into
/// `symbol_clock`, which lets them run CLOCK_TRACKING's causal rule instead
pub const CLOCK_TRACKING_RESLICE: i32 = 2;
pub const CLOCK_CAUSAL_OK: i32 = 0x100;
/// a device-side wait gave up (p25fe_shard_head_check: the head segment of a time shard never arrived)
pub const ERR_TIMEOUT: c_int = -8;
pub const SPECIALIZE_AUTO: i32 = 0;
pub const SPECIALIZE_OFF: i32 = -1;
pub const SPECIALIZE_REQUIRE: i32 = 1;
pub const SPECIALIZE_FORCE: i32 = 2;
pub const VARIANT_BUILTIN: c_int = 0;
pub const VARIANT_SPECIALIZED: c_int = 1;
pub const VARIANT_GENERIC: c_int = 2;
pub const GATHER_NONE: c_int = 0;
pub const GATHER_ROOT: c_int = 1;
pub const GATHER_ALL: c_int = 2;
pub const GATHER_ROOT_EXACT: c_int = 3;

/// p25fe_config_t
#[repr(C)]
#[derive(Clone, Copy)]
pub struct Config {
    pub abi_version: i32,
    pub device: i32,
    pub n_channels: i32,
    pub n_decim_taps: i32,
    pub n_chan_taps: i32,
    pub decim_taps: [f32; MAX_TAPS],   // p25_filts::DecimFir, src/demod.rs:27: tap 0 multiplies the newest sample
    pub chan_taps: [f32; MAX_TAPS],    // p25_filts::BandpassFir, src/demod.rs:29
    pub symbol_clock: i32,
    pub specialize: i32,
    pub fm_deviation_hz: u32,          // FmDemod::new(5000, BASEBAND_SAMPLE_RATE), src/demod.rs:54
    pub fm_sample_rate_hz: u32,
    pub fm_gain: f32,                  // 0: derived from the two above
    pub u8_scale: f32,                 // rtlsdr_iq::IQ as fma(b, scale, offset), src/demod.rs:83 ...
    pub u8_offset: f32,
    pub u8_lut_valid: i32,             // ... or as the table itself
    pub u8_lut: [f32; 256],
    pub decim_phase: i32,              // Decimator::new(5), src/demod.rs:50, 87-90: output m comes from input 5 m + decim_phase (4)
    pub n_avg_taps: i32,               // MovingAverage::new(10), src/demod.rs:52, 114, as a table (ten equal taps of 0.1)
    pub avg_taps: [f32; MAX_TAPS],
}

/// p25fe_anchor_t
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct Anchor {
    pub s: i64,
    pub hi: f32,
    pub mid: f32,
    pub lo: f32,
    pub valid: i32,                    // test `!= 0`; hand on unchanged (upper bits: docs/SPEC.md 3.8b)
    pub period_d: i32,
    pub period_n: i32,
}

/// p25fe_result_t
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ResultRec {
    pub n_baseband: u64,
    pub n_dibits: u64,
    pub n_sync: u64,
    pub anchor_out: Anchor,
    pub first_event: i64,
    pub n_dibits_after_first: u64,
    pub carry_end: i64,
    pub first_seg_end: i64,
    pub flags: u32,
    pub reserved: u32,
}

/// p25fe_nid_t
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct Nid {
    pub raw: u64,
    pub sync_pos: i64,
    pub nac: u16,
    pub duid: u8,
    pub n_errors: u8,
    pub valid: i32,
}

/// p25fe_code_stats_t / p25fe_chan_stats_t
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct CodeStats {
    pub words: u64,
    pub errs: u64,
    pub fixed: u64,
    pub size: u32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ChanStats {
    pub sig_power_dbm: f32,
    pub locked: i32,
    pub n_dibits: u64,
    pub n_sync: u64,
    pub last_sync_pos: i64,
    pub bch: CodeStats,
}

/// p25fe_windows_stats_t
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct WindowsStats {
    pub n_windows: u64,
    pub ms_total: f64,
    pub ms_h2d: f64,
    pub ms_compute: f64,
    pub pinned_input: i32,
    pub reserved: i32,
}

pub enum Handle {}
pub enum Shard {}

#[link(name = "p25fe")]
extern "C" {
    pub fn p25fe_default_config(cfg: *mut Config);
    pub fn p25fe_create(cfg: *const Config, out: *mut *mut Handle) -> c_int;
    pub fn p25fe_destroy(h: *mut Handle);
    pub fn p25fe_strerror(status: c_int) -> *const c_char;
    pub fn p25fe_last_hip_error(h: *const Handle) -> c_int;
    pub fn p25fe_device(h: *const Handle) -> c_int;
    pub fn p25fe_kernel_variant(h: *const Handle) -> c_int;
    pub fn p25fe_specialize(cfg: *const Config, dir: *const c_char, path_out: *mut c_char, path_cap: usize) -> c_int;
    pub fn p25fe_probe_variant(cfg: *const Config) -> c_int;
    pub fn p25fe_specialize_log(buf: *mut c_char, cap: usize) -> usize;
    // streaming, host buffers: the bodies of DemodTask::run (src/demod.rs:70-117) and RecvTask::run (src/recv.rs:148-150)
    pub fn p25fe_demod_u8(h: *mut Handle, iq: *const u8, n_bytes: usize, bb: *mut f32, bb_cap: usize, n_out: *mut usize,
                          power_dbm: *mut f32) -> c_int;
    pub fn p25fe_demod_cf32(h: *mut Handle, iq: *const f32, n_samples: usize, bb: *mut f32, bb_cap: usize, n_out: *mut usize,
                            power_dbm: *mut f32) -> c_int;
    pub fn p25fe_slice(h: *mut Handle, bb: *const f32, n: usize, dibits: *mut u8, cap: usize, n_dibits: *mut usize,
                       sync_pos: *mut i64, sync_dibit: *mut u64, sync_cap: usize, n_sync: *mut usize) -> c_int;
    pub fn p25fe_run_u8(h: *mut Handle, iq: *const u8, n_bytes: usize, dibits: *mut u8, cap: usize, n_dibits: *mut usize) -> c_int;
    pub fn p25fe_run_cf32(h: *mut Handle, iq: *const f32, n_samples: usize, dibits: *mut u8, cap: usize, n_dibits: *mut usize) -> c_int;
    pub fn p25fe_run_host_windows(h: *mut Handle, iq: *const c_void, fmt: c_int, n: usize, window: usize, dibits: *mut u8,
                                  cap: usize, n_dibits: *mut usize, stats: *mut WindowsStats) -> c_int;
    pub fn p25fe_resync(h: *mut Handle) -> c_int;
    pub fn p25fe_resync_at_dev(h: *mut Handle, d_idx: *const i64, n_idx: usize, idx_stride: usize) -> c_int;
    pub fn p25fe_reset(h: *mut Handle) -> c_int;
    pub fn p25fe_state_size(h: *const Handle, n: *mut usize) -> c_int;
    pub fn p25fe_state_export(h: *const Handle, buf: *mut c_void, cap: usize, n: *mut usize) -> c_int;
    pub fn p25fe_state_import(h: *mut Handle, buf: *const c_void, n: usize) -> c_int;
    // device-resident ranges (all pointers device pointers; stream = hipStream_t or null)
    pub fn p25fe_demod_dev(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize, abs0: u64,
                           d_bb: *mut f32, bb_stride: usize, d_power_dbm: *mut f32, stream: *mut c_void) -> c_int;
    pub fn p25fe_predecim_dev(h: *mut Handle, d_iq: *const f32, ch_stride: usize, n_hist: usize, n: usize, abs0: u64,
                              d_out: *mut f32, out_stride: usize, stream: *mut c_void) -> c_int;
    pub fn p25fe_n_predecim(abs0: u64, n: usize) -> usize;
    pub fn p25fe_n_baseband(abs0: u64, n: usize) -> usize;
    pub fn p25fe_slice_dev(h: *mut Handle, d_bb: *const f32, bb_stride: usize, n_hist_bb: usize, n_bb: usize, abs_bb0: u64,
                           d_anchor_in: *const Anchor, d_dibits: *mut u8, dibit_stride: usize, d_sync_pos: *mut i64,
                           d_sync_dibit: *mut u64, sync_stride: usize, d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_run_dev(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n: usize, d_dibits: *mut u8,
                         dibit_stride: usize, d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_run_dev_pipelined(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n: usize, d_dibits: *mut u8,
                                   dibit_stride: usize, d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_join_dev(h: *mut Handle, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_halo() -> usize;
    pub fn p25fe_shard_pass1(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize, abs0: u64,
                             d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_pass1_main(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize,
                                  abs0: u64, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_pass1_finish(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize,
                                    abs0: u64, d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_pass1_head(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize,
                                  abs0: u64, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_pipe_begin(h: *mut Handle, stream: *mut c_void, rx_stream: *mut *mut c_void) -> c_int;
    pub fn p25fe_shard_pipe_end(h: *mut Handle, last_stream: *mut c_void) -> c_int;
    pub fn p25fe_rx_stream(h: *mut Handle, rx_stream: *mut *mut c_void) -> c_int;
    pub fn p25fe_shard_head_check(h: *mut Handle) -> c_int;
    pub fn p25fe_shard_pass1_k1(h: *mut Handle, d_iq: *const c_void, fmt: c_int, ch_stride: usize, n_hist: usize, n: usize,
                                abs0: u64, stream: *mut c_void) -> c_int;
    pub fn p25fe_streams_share_queue(h: *mut Handle, stream_a: *mut c_void, stream_b: *mut c_void, shared: *mut c_int) -> c_int;
    pub fn p25fe_shard_pass2(h: *mut Handle, d_anchor_in: *const Anchor, d_dibits: *mut u8, dibit_stride: usize,
                             d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_pass2_dev(h: *mut Handle, d_summaries: *const ResultRec, d_shard_bb0: *const u64, d_shard_bb_n: *const u64,
                                 n_shards: usize, rank: usize, d_anchor_in: *mut Anchor, d_dibit_offset: *mut u64, d_dibits: *mut u8,
                                 dibit_stride: usize, d_dibits_dup: *mut u8, d_result: *mut ResultRec, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_resolve(summaries: *const ResultRec, shard_bb0: *const u64, shard_bb_n: *const u64, n_shards: usize,
                               symbol_clock: c_int, anchor_in: *mut Anchor, dibit_offset: *mut u64) -> c_int;
    pub fn p25fe_shard_resolve_dev(h: *mut Handle, d_summaries: *const ResultRec, d_shard_bb0: *const u64, d_shard_bb_n: *const u64,
                                   n_shards: usize, d_anchor_in: *mut Anchor, d_dibit_offset: *mut u64, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_compact_dev(h: *mut Handle, d_gathered: *const u8, cap: usize, d_dibit_offset: *const u64, n_shards: usize,
                                   d_out: *mut u8, out_cap: usize, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_compact_from_dev(h: *mut Handle, d_gathered: *const u8, cap: usize, d_dibit_offset: *const u64,
                                        first_shard: usize, n_shards: usize, d_out: *mut u8, out_cap: usize,
                                        stream: *mut c_void) -> c_int;
    pub fn p25fe_nid_dev(h: *mut Handle, d_dibits: *const u8, n_dibits: usize, d_sync_dibit: *const u64, d_sync_pos: *const i64,
                         n_sync: usize, d_out: *mut Nid, stream: *mut c_void) -> c_int;
    pub fn p25fe_nid(h: *mut Handle, dibits: *const u8, n_dibits: usize, sync_dibit: *const u64, sync_pos: *const i64, n_sync: usize,
                     out: *mut Nid) -> c_int;
    pub fn p25fe_nid_batch_dev(h: *mut Handle, d_dibits: *const u8, dibit_stride: usize, d_result: *const ResultRec,
                               d_sync_dibit: *const u64, d_sync_pos: *const i64, sync_stride: usize, d_out: *mut Nid,
                               stream: *mut c_void) -> c_int;
    pub fn p25fe_chan_stats_dev(h: *mut Handle, d_result: *const ResultRec, d_nid: *const Nid, sync_stride: usize,
                                d_power_dbm: *const f32, d_stats: *mut ChanStats, stream: *mut c_void) -> c_int;
    pub fn p25fe_channelise_dev(h: *mut Handle, d_iq: *const f32, n_hist: usize, n: usize, abs0: u64, d_out: *mut f32,
                                out_stride: usize, stream: *mut c_void) -> c_int;
    pub fn p25fe_n_baseband_h(h: *const Handle, abs0: u64, n: usize) -> usize;
    pub fn p25fe_profile_enable(h: *mut Handle, on: c_int) -> c_int;
    pub fn p25fe_profile_read(h: *mut Handle, ms: *mut f64, n_calls: *mut u64) -> c_int;
}

#[link(name = "p25fe_rccl")]
extern "C" {
    pub fn p25fe_rccl_unique_id(id128: *mut u8) -> c_int;
    pub fn p25fe_shard_create(h: *mut Handle, rank: c_int, world: c_int, id128: *const u8, n_per_rank: usize, out: *mut *mut Shard) -> c_int;
    pub fn p25fe_shard_destroy(s: *mut Shard);
    pub fn p25fe_shard_dibit_cap(s: *const Shard) -> usize;
    pub fn p25fe_shard_step(s: *mut Shard, d_buf: *mut c_void, fmt: c_int, d_dibits: *mut u8, d_result: *mut ResultRec, gather: c_int,
                            stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_step_pipelined(s: *mut Shard, d_buf: *mut c_void, fmt: c_int, d_dibits: *mut u8, d_result: *mut ResultRec,
                                      gather: c_int, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_join(s: *mut Shard, stream: *mut c_void) -> c_int;
    pub fn p25fe_shard_offsets(s: *mut Shard, offsets: *mut u64) -> c_int;
    pub fn p25fe_shard_stream_dev(s: *const Shard) -> *const u8;
    pub fn p25fe_shard_comm_ms(s: *mut Shard, ms: *mut f64, n_steps: *mut u64) -> c_int;
    pub fn p25fe_shard_comm_timing(s: *mut Shard, every: c_int) -> c_int;
    pub fn p25fe_shard_gather_ran(s: *const Shard) -> c_int;
    pub fn p25fe_shard_info(s: *const Shard, out: *mut ShardInfo) -> c_int;
    pub fn p25fe_shard_prepare(s: *mut Shard, stream: *mut c_void) -> c_int;
}

/// `p25fe_shard_info_t`: what the shard object itself knows about the job (RCCL's own rank count, this rank's GPU, the agreed layout)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ShardInfo {
    pub rank: i32,
    pub world: i32,
    pub rccl_ranks: i32,
    pub rccl_rank: i32,
    pub device: i32,
    pub comms: i32,
    pub pipe_layout: i32,
    pub gather_ran: i32,
    pub staged: i32,
    pub head_wait: i32,
    pub broken: i32,
    pub reserved: i32,
    pub steps: u64,
    pub pci_bus_id: [c_char; 32],
}

/// Owning wrapper of one handle.  One per thread, like the tasks of the reference (src/main.rs:270-287).
pub struct FrontEnd {
    h: *mut Handle,
}
unsafe impl Send for FrontEnd {}

fn check(rc: c_int, what: &str) {
    // the reference's convention for this path: every failure aborts (`expect`, panic = "abort", Cargo.toml:50-51)
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(p25fe_strerror(rc)) };
        panic!("{}: {}", what, msg.to_string_lossy());
    }
}

impl FrontEnd {
    /// The build's own numbers (decimation 5, boxcar 10, 5 kHz at 48 kHz: src/demod.rs:50-54).
    pub fn new(device: i32) -> Self {
        let mut cfg: Config = unsafe { std::mem::zeroed() };
        unsafe { p25fe_default_config(&mut cfg) };
        cfg.device = device;
        Self::with_config(&cfg)
    }

    /// The reference's own numbers: p25_filts' tables, rtlsdr_iq's table, FmDemod's arguments.  The library compiles its
    /// kernels for them (once per set of numbers, cached on disk): no slower than the built-in ones.
    pub fn with_tables(device: i32, decim: &[f32], chan: &[f32], iq_lut: Option<&[f32; 256]>, deviation: u32, rate: u32) -> Self {
        let mut cfg: Config = unsafe { std::mem::zeroed() };
        unsafe { p25fe_default_config(&mut cfg) };
        cfg.device = device;
        cfg.n_decim_taps = decim.len() as i32;
        cfg.n_chan_taps = chan.len() as i32;
        cfg.decim_taps[..decim.len()].copy_from_slice(decim);
        cfg.chan_taps[..chan.len()].copy_from_slice(chan);
        if let Some(lut) = iq_lut {
            cfg.u8_lut_valid = 1;
            cfg.u8_lut = *lut;
        }
        cfg.fm_deviation_hz = deviation;
        cfg.fm_sample_rate_hz = rate;
        Self::with_config(&cfg)
    }

    pub fn with_config(cfg: &Config) -> Self {
        let mut h: *mut Handle = std::ptr::null_mut();
        check(unsafe { p25fe_create(cfg, &mut h) }, "unable to create the front end");
        FrontEnd { h: h }
    }

    /// One loop body of DemodTask::run (src/demod.rs:70-117); returns the number of baseband samples written.
    pub fn demod_u8(&mut self, bytes: &[u8], baseband: &mut [f32], power_dbm: Option<&mut f32>) -> usize {
        let mut n_out = 0usize;
        let p = power_dbm.map_or(std::ptr::null_mut(), |r| r as *mut f32);
        check(unsafe { p25fe_demod_u8(self.h, bytes.as_ptr(), bytes.len(), baseband.as_mut_ptr(), baseband.len(), &mut n_out, p) },
              "unable to demodulate");
        n_out
    }

    /// The sample loop of RecvTask::run (src/recv.rs:148-150) down to "a dibit exists"; returns (dibits, sync events).
    pub fn slice(&mut self, baseband: &[f32], dibits: &mut [u8], sync_pos: &mut [i64], sync_dibit: &mut [u64]) -> (usize, usize) {
        let (mut nd, mut ns) = (0usize, 0usize);
        let cap = sync_pos.len().min(sync_dibit.len());
        check(unsafe { p25fe_slice(self.h, baseband.as_ptr(), baseband.len(), dibits.as_mut_ptr(), dibits.len(), &mut nd,
                                   sync_pos.as_mut_ptr(), sync_dibit.as_mut_ptr(), cap, &mut ns) }, "unable to slice");
        (nd, ns.min(cap))
    }

    /// MessageReceiver::resync (src/recv.rs:136, 179)
    pub fn resync(&mut self) {
        check(unsafe { p25fe_resync(self.h) }, "unable to resync");
    }

    /// A long recording from host memory at bus speed (the replay path, src/replay.rs:26-37, for IQ files).
    pub fn run_host_windows_u8(&mut self, bytes: &[u8], dibits: &mut [u8]) -> (usize, WindowsStats) {
        let mut nd = 0usize;
        let mut st = WindowsStats::default();
        check(unsafe { p25fe_run_host_windows(self.h, bytes.as_ptr() as *const c_void, FMT_U8, bytes.len() / 2, 0, dibits.as_mut_ptr(),
                                              dibits.len(), &mut nd, &mut st) }, "unable to run the capture");
        (nd, st)
    }

    pub fn raw(&self) -> *mut Handle {
        self.h
    }
}

impl Drop for FrontEnd {
    fn drop(&mut self) {
        unsafe { p25fe_destroy(self.h) };
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DemodTask with the handle in place of decim / bandpass / avg / demod (src/demod.rs:25-40): same constructor arguments, same
// channels, same messages.  Replaces the struct and impl of src/demod.rs:24-119; `power_dbm` (src/demod.rs:123-134) stays
// for other callers.  (`use` lines as in src/demod.rs:3-21 minus the six DSP crates.)
// ---------------------------------------------------------------------------------------------------------------------
/*
pub struct DemodTask {
    fe: FrontEnd,
    reader: Receiver<Checkout<Vec<u8>>>,
    hub: mio_more::channel::Sender<HubEvent>,
    chan: Sender<RecvEvent>,
}

impl DemodTask {
    pub fn new(reader: Receiver<Checkout<Vec<u8>>>, hub: mio_more::channel::Sender<HubEvent>, chan: Sender<RecvEvent>) -> Self {
        DemodTask {
            // the reference's own tables and constants, as data (src/demod.rs:27-29, 54, 83)
            fe: FrontEnd::with_tables(0, &DECIM_TAPS, &BANDPASS_TAPS, Some(&IQ_BYTE_TABLE), 5000, BASEBAND_SAMPLE_RATE),
            reader: reader, hub: hub, chan: chan,
        }
    }

    pub fn run(&mut self) {
        let mut pool = Pool::with_capacity(16, || vec![0.0; BUF_SAMPLES]);                 // src/demod.rs:63
        let mut notifier = Throttler::new(4);                                               // src/demod.rs:67
        loop {
            let bytes = self.reader.recv().expect("unable to receive sdr samples");        // src/demod.rs:70
            let mut baseband = pool.checkout().expect("unable to allocate baseband");      // src/demod.rs:103
            unsafe { baseband.set_len(BUF_SAMPLES); }
            let mut power = 0f32;
            let mut want = false;
            notifier.throttle(|| want = true);                                              // src/demod.rs:95
            let n = self.fe.demod_u8(&bytes[..], &mut baseband[..], if want { Some(&mut power) } else { None });
            unsafe { baseband.set_len(n); }                                                 // 3276 / 3277, src/demod.rs:87-90, 106
            if want {
                self.hub.send(HubEvent::UpdateSignalPower(power)).expect("unable to send signal power");   // src/demod.rs:99
            }
            self.chan.send(RecvEvent::Baseband(baseband)).expect("unable to send baseband");               // src/demod.rs:116
        }
    }
}
*/
