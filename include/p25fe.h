/*
 * p25fe.h -- C ABI of the MI355X-native P25 front end (IQ -> 48 kHz baseband -> C4FM dibits).
 *
 * Drop-in boundary for the hot path of kchmck/p25rx.  The reference has no FFI for this path:
 * it is two Rust task types wired by channels.  Each entry point below names
 * the reference item whose work it replaces; INTEGRATION.md shows the Rust `extern "C"` block
 * and the DemodTask / RecvTask edits a maintainer would make.
 *
 *   reference item (file:line in /root/reference)                  replaced by
 *   -------------------------------------------------------------  ---------------------------
 *   DemodTask::new   src/demod.rs:44-59                             p25fe_create
 *     Decimator::new(5)                 :50, 87-90                  p25fe_config_t.decim_phase (the factor 5 is the build's)
 *     MovingAverage::new(10)            :52, 114                    p25fe_config_t.n_avg_taps / avg_taps
 *     DecimFir / BandpassFir tables     :27-29                      p25fe_config_t.decim_taps / chan_taps
 *     FmDemod::new(5000, 48000)         :54                         p25fe_config_t.fm_deviation_hz / fm_sample_rate_hz / fm_gain
 *     rtlsdr_iq::IQ                     :83                         p25fe_config_t.u8_scale / u8_offset / u8_lut
 *     (type-level FIRs = compile-time tables)                       p25fe_specialize / p25fe_kernel_variant
 *   DemodTask::run   src/demod.rs:70-117 loop body, u8 chunk        p25fe_demod_u8
 *     IQ[s] LUT                         :74-84
 *     decim.decim_in_place              :87-90
 *     bandpass.feed                     :93
 *     power_dbm                         :95-101, :123-134           (power_dbm out-parameter)
 *     demod.feed / avg.feed             :109-114
 *   RecvTask::run sample loop  src/recv.rs:148-150, 204-210         p25fe_slice
 *     msg.feed(s) down to "a dibit exists" (p25 crate, un-vendored)
 *   MessageReceiver::resync    src/recv.rs:136, 179                 p25fe_resync
 *   reader -> pool -> demodulator pipeline  src/sdr.rs:25-33,        p25fe_run_host_windows (a long host capture as a
 *     src/demod.rs:62-70, 103                                         pipeline of windows at the speed of the bus)
 *   (state hand-off for time-sharded captures; no reference item)   p25fe_state_export/import,
 *                                                                   p25fe_shard_*
 *
 * Conventions kept from the reference: streaming semantics -- any chunking of the input gives
 * the same concatenated output, state lives in the handle (src/demod.rs:25-40); the output count
 * is data dependent (src/demod.rs:87-90).  Convention NOT kept: the reference aborts on every
 * error (`expect`, panic = "abort", Cargo.toml:50-51); here every call returns 0 or a negative
 * p25fe_status, because unwinding across extern "C" is undefined.
 *
 * Threading: one handle per thread (the reference moves each task into its own thread,
 * src/main.rs:270-287).  No globals, no callbacks.  Host-pointer calls are synchronous.
 * *_dev calls take device pointers, enqueue on `stream` (a hipStream_t passed as void*) and do
 * not synchronise.  There is no CPU fallback: without a HIP device p25fe_create fails.
 */
#ifndef P25FE_H
#define P25FE_H

#ifndef __HIPCC_RTC__      /* (the library's own kernel source includes this header under hipRTC, which has no system headers) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define P25FE_MAX_TAPS 64
#define P25FE_ABI_VERSION 6        /* 3: symbol_clock in the config, clock period in p25fe_anchor_t, carry_end / first_seg_end /
                                      flags in p25fe_result_t, p25fe_resync_at_dev, symbol_clock argument of p25fe_shard_resolve
                                      4: the run-time arguments of the reference's constructors in the config (FM deviation / rate,
                                      the u8 -> float LUT), `specialize`, p25fe_specialize / p25fe_kernel_variant,
                                      p25fe_run_host_windows
                                      5: the LAST constructor numbers: the post-discriminator filter as a table (avg_taps:
                                      MovingAverage::new(10)) and the decimator's phase (decim_phase: Decimator::new(5));
                                      p25fe_n_baseband_h; p25fe_shard_halo() is 2 560; p25fe_shard_pass1_head / _pass2_dev /
                                      _compact_from_dev; p25fe_probe_variant
                                      6: symbol_clock = 2 is refused by the calls that cannot run SPEC 3.8c unless P25FE_CLOCK_CAUSAL_OK
                                      is or-ed in; P25FE_ERR_TIMEOUT, p25fe_shard_head_check (the detection's wait for a shard's head is
                                      bounded); p25fe_rx_stream; p25fe_rccl.h: p25fe_shard_info, p25fe_shard_prepare (no step probes or
                                      synchronises streams any more), the communicator layout agreed by all ranks */

typedef enum p25fe_status {
    P25FE_OK = 0,
    P25FE_ERR_ARG = -1,        /* null pointer, bad size, bad config */
    P25FE_ERR_NO_DEVICE = -2,  /* no HIP device / wrong architecture */
    P25FE_ERR_HIP = -3,        /* a HIP runtime call failed (see p25fe_last_hip_error) */
    P25FE_ERR_CAPACITY = -4,   /* output buffer too small; nothing consumed */
    P25FE_ERR_FORMAT = -5,     /* u8 / cf32 mixed within one stream */
    P25FE_ERR_NOMEM = -6,
    P25FE_ERR_JIT = -7,        /* specialising the kernels failed (p25fe_specialize_log has the compiler's words) */
    P25FE_ERR_TIMEOUT = -8     /* a device-side wait gave up (the head segment of a time shard never arrived: p25fe_shard_head_check) */
} p25fe_status;

typedef enum p25fe_format {
    P25FE_FMT_CF32 = 0,        /* interleaved float32 I,Q (num::Complex32 layout) */
    P25FE_FMT_U8 = 1           /* interleaved uint8 I,Q (RTL-SDR; src/demod.rs:74-76) */
} p25fe_format;

/* Replaces the compile-time DSP parameters of DemodTask::new (src/demod.rs:49-54: every constructor argument of the four DSP
 * objects, ABI 5), the type-level tap tables of p25_filts (DecimFir / BandpassFir, src/demod.rs:27-29), the arguments of
 * FmDemod::new (src/demod.rs:54) and the rtlsdr_iq::IQ lookup table (src/demod.rs:83).  Up to P25FE_MAX_TAPS = 64 taps per filter; a table of n taps is evaluated as 31 / 41 taps (64 / 64
 * as soon as either table is longer) with zero coefficients at the old end: the same filter on finite samples; a NaN / Inf
 * sample reaches that many taps' worth of outputs (docs/SPEC.md 3.3).
 *
 * Which kernels run (p25fe_kernel_variant): the build's own numbers (p25fe_default_config) run as the immediate-coefficient
 * kernels compiled into the library.  ANY other table / constant set is compiled the same way -- the reference fixes its
 * tables at compile time too (type-level FIRs, src/demod.rs:27-29) -- by hipRTC at p25fe_create (or ahead of time by
 * p25fe_specialize) from the kernel source embedded in the library, cached on disk under a hash of the numbers; the
 * generic kernels (taps broadcast-read from LDS, the LUT in LDS) are the fallback when that is switched off or fails. */
typedef struct p25fe_config {
    int32_t abi_version;                 /* P25FE_ABI_VERSION */
    int32_t device;                      /* HIP device ordinal */
    int32_t n_channels;                  /* independent channels per call, channel-major */
    int32_t n_decim_taps;
    int32_t n_chan_taps;
    float decim_taps[P25FE_MAX_TAPS];    /* 240k -> 48k anti-alias, tap 0 multiplies the newest sample */
    float chan_taps[P25FE_MAX_TAPS];     /* 48 kHz channel-select low-pass */
    int32_t symbol_clock;                /* P25FE_CLOCK_FIXED (0, default): symbol instants at a fixed 10-sample stride from the last
                                            sync word, the receiver structure of the reference; P25FE_CLOCK_TRACKING (1): docs/SPEC.md
                                            3.8b -- the stride is the measured interval between the last two sync words over its
                                            symbol count, instants are read by 4-tap interpolation, the receiver runs 2 samples
                                            behind the baseband; P25FE_CLOCK_TRACKING_RESLICE (2): the same, and the calls that hold
                                            a whole range re-slice the first frame of a lock run (SPEC 3.8c, see the define; the calls
                                            that cannot are refused unless P25FE_CLOCK_CAUSAL_OK is or-ed in) */
    int32_t specialize;                  /* P25FE_SPECIALIZE_AUTO (0): non-default numbers get immediate-coefficient kernels (cache, then
                                            hipRTC), the generic kernels if that fails; _OFF (-1): always the generic kernels for
                                            non-default numbers; _REQUIRE (1): p25fe_create fails with P25FE_ERR_JIT instead of
                                            falling back; _FORCE (2): as _REQUIRE, and the build's own numbers are specialised too
                                            (measurement: the hipRTC product beside the library's own code object) */
    /* ---- ABI 4 ---- */
    uint32_t fm_deviation_hz;            /* FmDemod::new(5000, BASEBAND_SAMPLE_RATE), src/demod.rs:54: first argument */
    uint32_t fm_sample_rate_hz;          /* ... second argument: the rate of the discriminator's input (src/consts.rs:13: 48 000) */
    float fm_gain;                       /* 0 (default): the discriminator's output scale is (float)(fm_sample_rate_hz / (2 pi
                                            fm_deviation_hz)), evaluated in double and rounded once (docs/SPEC.md 3.4); any other
                                            finite value is used verbatim (a dump of demod_fm's own constant, tools/pin/) */
    float u8_scale, u8_offset;           /* rtlsdr_iq::IQ (src/demod.rs:83) as arithmetic: byte b -> fma((float)b, u8_scale, u8_offset)
                                            (docs/SPEC.md 3.1; defaults 2 / 255, -1) */
    int32_t u8_lut_valid;                /* non-zero: u8_lut[b] IS the value of byte b (I and Q alike) and u8_scale / u8_offset are
                                            ignored.  A table that equals fma(b, s, o) bit for bit for some (s, o) the library
                                            finds runs as arithmetic; any other table is looked up in LDS. */
    float u8_lut[256];
    /* ---- ABI 5 ---- */
    int32_t decim_phase;                 /* Decimator::new(5) / decim_in_place (src/demod.rs:50, 87-90): output m of the 5:1 decimator
                                            is produced by the input sample with absolute index 5 m + decim_phase, 0..4.  Default 4
                                            (the 5th, 10th, ... sample: 16 384-sample chunks give 3 276, 3 277, 3 277, ... outputs). */
    int32_t n_avg_taps;                  /* MovingAverage::new(10) (src/demod.rs:52, 114) as a table: the real FIR behind the
                                            discriminator, 1 .. P25FE_MAX_TAPS taps, tap 0 on the newest sample (docs/SPEC.md 3.5).
                                            A table whose taps are ALL EQUAL is a moving average and is evaluated as one: the
                                            samples summed newest first, then one multiply by the tap -- the default, ten taps of
                                            (float)0.1, gives the bits of every earlier ABI.  Any other table (e.g. the raised
                                            cosine P25FE_RC_AVG_TAPS of p25fe_spec.h) is evaluated like the other two FIRs: one
                                            accumulator from +0, fma in tap order. */
    float avg_taps[P25FE_MAX_TAPS];
} p25fe_config_t;

#define P25FE_CLOCK_FIXED 0
#define P25FE_CLOCK_TRACKING 1
/* P25FE_CLOCK_TRACKING plus docs/SPEC.md 3.8c in the calls that hold a whole range in device memory (p25fe_run_dev,
 * p25fe_run_dev_pipelined, p25fe_slice_dev): a detection without a period of its own -- the first of a lock run -- is sliced with
 * the period of the interval that STARTS at it, once the next sync word of the range confirms one, and every detection's instants
 * start from its refined position s + f / 4 instead of the whole sample s (0 symbol errors of 2.88 M at 150 ppm where
 * P25FE_CLOCK_TRACKING leaves 29 and the fixed stride 16 502).  That is not causal, so the calls that see the stream in pieces --
 * p25fe_slice, p25fe_run_u8 / _cf32, p25fe_run_host_windows, the time-shard passes (and with them p25fe_shard_step) -- cannot run it: they
 * have P25FE_CLOCK_TRACKING's rule only ("any chunking gives the same output", src/demod.rs:25-40).  ABI 6: on a handle created with
 * symbol_clock = 2 those calls return P25FE_ERR_ARG -- a request for one receiver is not answered with another's dibits -- unless the
 * handle was created with symbol_clock = P25FE_CLOCK_TRACKING_RESLICE | P25FE_CLOCK_CAUSAL_OK: 3.8c where a call holds the range, 3.8b
 * everywhere else, and a resident call and a streaming call then differ, by design (what ABI 5's mode 2 did without being asked). */
#define P25FE_CLOCK_TRACKING_RESLICE 2
#define P25FE_CLOCK_CAUSAL_OK 0x100
#define P25FE_SPECIALIZE_AUTO 0
#define P25FE_SPECIALIZE_OFF (-1)
#define P25FE_SPECIALIZE_REQUIRE 1
#define P25FE_SPECIALIZE_FORCE 2

typedef struct p25fe p25fe_t;

/* Symbol-timing anchor carried between calls / shards: the last frame-sync detection. */
typedef struct p25fe_anchor {
    int64_t s;                           /* absolute baseband index of the sync word's last symbol */
    float hi, mid, lo;                   /* slicer thresholds derived from that sync word */
    int32_t valid;                       /* 0: no lock.  Non-zero: locked -- bit 0 set; with the tracking clock bits 8..10 carry the sync
                                            position's fraction in quarter samples (3-bit two's complement, docs/SPEC.md 3.8b: it
                                            enters the NEXT detection's period; 0 from the fixed-stride receiver).  Test `valid != 0`. */
    int32_t period_d, period_n;          /* symbol period period_d / period_n samples (10 / 1 unless the clock tracks -- then quarter
                                            samples over four times the interval's symbol count; 0 / 0 reads as 10 / 1) */
} p25fe_anchor_t;

/* Per-channel summary of one processed range (device or host memory, see each call). */
typedef struct p25fe_result {
    uint64_t n_baseband;                 /* baseband samples produced */
    uint64_t n_dibits;                   /* dibits produced */
    uint64_t n_sync;                     /* frame-sync detections */
    p25fe_anchor_t anchor_out;           /* anchor after the range */
    int64_t first_event;                 /* baseband index at which the range's first own detection is decided (s + W), -1 if none */
    uint64_t n_dibits_after_first;       /* dibits governed by the range's own detections */
    /* what p25fe_shard_resolve needs beyond that when the clock tracks or lock is dropped inside ranges: */
    int64_t carry_end;                   /* index from which the carry-in anchor no longer governs (first_event + 1, or the first lock
                                            drop if that comes first; never negative); -1: it governs the whole range */
    int64_t first_seg_end;               /* end (exclusive) of the interval the first own detection governs: the next event of the range (detection or lock drop) or the range's end */
    uint32_t flags;                      /* P25FE_RES_* */
    uint32_t reserved;                   /* tracking clock: quarter-sample fraction (3-bit two's complement) of the first own detection's sync position; else 0 */
} p25fe_result_t;
#define P25FE_RES_FIRST_TRACKS_CARRY 1u  /* no lock drop between the range's start and its first detection: with a tracking clock that
                                            detection takes its period from the carry-in (n_dibits_after_first assumed 10 / 1) */
#define P25FE_RES_OUT_PERIOD_FROM_CARRY 2u /* anchor_out's period is that same interval (the range has one detection, nothing else) */

void p25fe_default_config(p25fe_config_t *cfg);
int p25fe_create(const p25fe_config_t *cfg, p25fe_t **out);
void p25fe_destroy(p25fe_t *h);
const char *p25fe_strerror(int status);
int p25fe_last_hip_error(const p25fe_t *h);      /* raw hipError_t of the last P25FE_ERR_HIP */
int p25fe_device(const p25fe_t *h);              /* HIP device ordinal the handle lives on (p25fe_config_t.device); < 0: null handle */

/* Which front-end kernels the handle launches: the library's own immediate-coefficient kernels (the config's numbers are
 * the build's), kernels specialised for the config's numbers (cache / hipRTC), or the generic LDS-tap kernels. */
#define P25FE_VARIANT_BUILTIN 0
#define P25FE_VARIANT_SPECIALIZED 1
#define P25FE_VARIANT_GENERIC 2
int p25fe_kernel_variant(const p25fe_t *h);      /* < 0: null handle */
/* Ahead-of-time form of what p25fe_create does for non-default numbers (the `make SPEC=` step; needs no GPU): compile the
 * front-end kernels for cfg's tables and constants and store the code object as <dir>/p25fe-<hash>.hsaco (dir NULL: the
 * cache directory -- $P25FE_CACHE_DIR, else $XDG_CACHE_HOME/p25fe, else $HOME/.cache/p25fe, else /tmp/p25fe-cache-<uid>;
 * p25fe_create looks in $P25FE_SPEC_DIR first, then there; what IT compiles is stored as p25fe-<hash>-rtc<version>.hsaco and
 * looked for before the plain name, so another toolchain's object is never preferred over this one's).  path_out (nullable) receives the file name.  Returns P25FE_OK
 * (also when cfg holds the build's own numbers: nothing to do, path_out = ""), P25FE_ERR_JIT or P25FE_ERR_ARG. */
int p25fe_specialize(const p25fe_config_t *cfg, const char *dir, char *path_out, size_t path_cap);
/* Which kernels a handle made from cfg would run ON THIS HOST (P25FE_VARIANT_*), by the same steps p25fe_create takes --
 * $P25FE_SPEC_DIR, the cache, hipRTC (the object is stored in the cache), else the fallback -- but without a device: a
 * deployment check.  When P25FE_SPECIALIZE_AUTO ends on the generic kernels, p25fe_create and this call say so ONCE per
 * process on stderr (P25FE_QUIET=1 silences it).  P25FE_ERR_JIT if cfg->specialize demands what cannot be had.
 * Code objects are read from / written to directories that are real directories owned by the caller (or root) and
 * writable by their owner only; every file carries a trailer that ties its content to cfg's numbers and is verified
 * before it reaches the loader; files under $P25FE_SPEC_DIR are never deleted. */
int p25fe_probe_variant(const p25fe_config_t *cfg);
/* compiler log of the calling thread's last p25fe_specialize / p25fe_create; returns the length copied (NUL-terminated) */
size_t p25fe_specialize_log(char *buf, size_t cap);

/* ---- streaming, host buffers: the bodies of DemodTask::run and RecvTask::run ----------------
 * Multi-channel handles take channel-major buffers: channel c starts at c * (elements per
 * channel) of the call; every channel gets the same number of input samples. */

/* src/demod.rs:70-117 for one chunk of interleaved u8 I/Q.  n_bytes even.  Writes *n_out
 * baseband samples per channel to bb (channel c at bb + c * bb_cap).  power_dbm (nullable)
 * receives power_dbm() of the post-channel-filter chunk per channel (src/demod.rs:97); the
 * every-4th-chunk throttle (src/demod.rs:67, 95) is the caller's. */
int p25fe_demod_u8(p25fe_t *h, const uint8_t *iq, size_t n_bytes, float *bb, size_t bb_cap, size_t *n_out,
                   float *power_dbm);
/* Same for Complex32 input (BASELINE.json configs 2-5). */
int p25fe_demod_cf32(p25fe_t *h, const float *iq, size_t n_samples, float *bb, size_t bb_cap, size_t *n_out,
                     float *power_dbm);

/* src/recv.rs:148-150: feed n baseband samples per channel; one dibit per byte (0..3) to
 * dibits (channel c at dibits + c * cap), counts to n_dibits[c].  sync_pos / sync_dibit
 * (nullable) receive, per detection, the absolute baseband index of the sync word's last
 * symbol and the index (in the channel's dibit stream) of the first dibit it governs;
 * n_sync[c] gets the number of detections (may exceed sync_cap; extra ones are not stored).
 * Capacity: cap >= n / 10 + 1 is required (an undisturbed lock), cap = n / 6 + 2 is always enough (every re-anchor starts
 * a new symbol grid and two of them are at least 6 samples apart); a chunk that needs more than `cap` returns
 * P25FE_ERR_CAPACITY with the stream state untouched -- repeat the call with more room. */
int p25fe_slice(p25fe_t *h, const float *bb, size_t n, uint8_t *dibits, size_t cap, size_t *n_dibits,
                int64_t *sync_pos, uint64_t *sync_dibit, size_t sync_cap, size_t *n_sync);

/* Both halves with the baseband kept in HBM (DemodTask -> RecvTask without the channel hop).  Capacity as for
 * p25fe_slice with n = the chunk's baseband samples (p25fe_n_baseband): n_samples / 30 + 4 is always enough. */
int p25fe_run_u8(p25fe_t *h, const uint8_t *iq, size_t n_bytes, uint8_t *dibits, size_t cap, size_t *n_dibits);
int p25fe_run_cf32(p25fe_t *h, const float *iq, size_t n_samples, uint8_t *dibits, size_t cap, size_t *n_dibits);

/* A LONG capture in host memory through the same path at the speed of the bus: the capture is cut into windows of `window`
 * samples per channel (0: 64 MB worth; rounded down to a multiple of 8, at least 8 192); window k + 1 travels to the GPU
 * (hipMemcpyAsync on a copy stream, two device windows) while window k is demodulated and sliced, and window k - 1's dibits
 * travel back.  Filter and receiver state cross the window boundaries exactly as they cross a time shard's: the last
 * p25fe_shard_halo() samples stay in front of the next window (device-to-device), the anchor is handed on in device
 * memory.  The reference's shape is the same pipeline with a 16-deep buffer pool between reader and demodulator
 * (src/demod.rs:62-70, 103; src/sdr.rs:25-33).  Continues the handle's stream (any chunking of p25fe_run_* and this call
 * gives the same concatenated dibits) and leaves it ready for more.
 * iq: [C][n] channel-major.  Pinned memory (hipHostMalloc / hipHostRegister) is copied from directly; pageable memory is
 * staged through two pinned windows of the library by the calling thread.  dibits: [C][cap]; capacity as for p25fe_run_*.
 * stats (nullable): where the time went.
 * On an error (P25FE_ERR_CAPACITY from a window's count, P25FE_ERR_HIP) the handle's stream state has not moved -- the call
 * can be repeated -- and nothing of the call is in flight any more, but `dibits` may already hold the dibits of earlier
 * windows and n_dibits reads 0: treat the output buffers as undefined. */
typedef struct p25fe_windows_stats {
    uint64_t n_windows;
    double ms_total;                     /* wall time of the call */
    double ms_h2d;                       /* the H2D copies' own time (events on the copy stream), summed */
    double ms_compute;                   /* the windows' kernels (events on the compute stream), summed */
    int32_t pinned_input;                /* 1: copied straight from the caller's memory, 0: staged by the calling thread */
    int32_t reserved;
} p25fe_windows_stats_t;
int p25fe_run_host_windows(p25fe_t *h, const void *iq, int fmt, size_t n, size_t window, uint8_t *dibits, size_t cap,
                           size_t *n_dibits, p25fe_windows_stats_t *stats);

/* MessageReceiver::resync (src/recv.rs:136, 179): drop symbol lock at the current position. */
int p25fe_resync(p25fe_t *h);
/* The same for device-resident ranges, where "the current position" lies INSIDE the range (RecvTask handles
 * RecvEvent::SetControlFreq between two baseband chunks, src/recv.rs:127-137; a resident capture spans many): the NEXT
 * call on this handle that runs the receiver (p25fe_slice_dev, p25fe_run_dev, p25fe_run_dev_pipelined,
 * p25fe_shard_pass1 / _finish + the p25fe_shard_pass2 that follows; also the host-buffer calls p25fe_slice,
 * p25fe_run_u8 / _cf32, whose range is the chunk) drops lock before each listed sample -- exactly as
 * if resync() had been called between feeding samples q - 1 and q.  d_idx: device array, n_idx ascending ABSOLUTE
 * baseband indices per channel, channel c at d_idx + c * idx_stride (pad a shorter list with INT64_MAX); it must stay
 * valid and unchanged until that call's kernels have run.  n_idx = 0 cancels.  The list is consumed by that call -- a host-buffer
 * call that returns P25FE_ERR_CAPACITY has consumed nothing, the list included; p25fe_reset and p25fe_state_import drop it. */
int p25fe_resync_at_dev(p25fe_t *h, const int64_t *d_idx, size_t n_idx, size_t idx_stride);
/* Forget all stream state (a new DemodTask + MessageReceiver). */
int p25fe_reset(p25fe_t *h);

/* Filter histories, decimator phase, FM/boxcar context and symbol lock as an opaque blob. */
int p25fe_state_size(const p25fe_t *h, size_t *n);
int p25fe_state_export(const p25fe_t *h, void *buf, size_t cap, size_t *n);
int p25fe_state_import(p25fe_t *h, const void *buf, size_t n);

/* ---- device-resident ranges: the measured path -------------------------------------------------
 * d_iq points at the first OWNED sample of channel 0; channel c starts ch_stride elements
 * (complex samples for CF32, byte pairs for U8) later.  n_hist valid samples precede each
 * channel's first owned sample in memory (0 at the start of a stream: history reads as zero,
 * exactly like the zero-initialised filters of DemodTask::new); abs0 is the absolute index of
 * the first owned sample in its stream (fixes the 5:1 grid, src/demod.rs:87-90).
 * All outputs are device pointers; nothing is synchronised. */

/* stages 1-5: writes n_baseband(n, abs0) samples per channel to d_bb (+ c * bb_stride). */
int p25fe_demod_dev(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                    uint64_t abs0, float *d_bb, size_t bb_stride, float *d_power_dbm, void *stream);

/* Stage 0 of BASELINE.json config 3 (2.4 Msps front end; the reference is fixed at 240 ksps, src/consts.rs:11):
 * 10:1 decimating FIR, cf32 @ 2.4 Msps -> cf32 @ 240 ksps, P25FE_T0 taps of p25fe_spec.h.  Same range
 * conventions as p25fe_demod_dev; writes p25fe_n_predecim(abs0, n) complex samples per channel to d_out
 * (+ c * out_stride complex samples).  Feed d_out to p25fe_run_dev / p25fe_demod_dev. */
int p25fe_predecim_dev(p25fe_t *h, const float *d_iq, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                       float *d_out, size_t out_stride, void *stream);
size_t p25fe_n_predecim(uint64_t abs0, size_t n);

/* stages 6-7 on device baseband.  d_bb points at the first owned sample; n_hist_bb valid
 * samples precede it; abs_bb0 is its absolute index; d_anchor_in (nullable = no lock) is the
 * carry-in per channel.  d_result[c] is filled per channel.
 * dibit_stride (here and in every *_dev call) is the row stride AND the per-channel capacity of d_dibits: a receiver
 * that re-anchors on every sync word follows the transmitter's symbol clock, so a range of n baseband samples can hold
 * more than n / 10 dibits (n / 6 at most, under back-to-back detections).  d_result[c].n_dibits is always the exact
 * count; dibits past the capacity are not stored (n_dibits > dibit_stride tells the caller so). */
int p25fe_slice_dev(p25fe_t *h, const float *d_bb, size_t bb_stride, size_t n_hist_bb, size_t n_bb,
                    uint64_t abs_bb0, const p25fe_anchor_t *d_anchor_in, uint8_t *d_dibits, size_t dibit_stride,
                    int64_t *d_sync_pos, uint64_t *d_sync_dibit, size_t sync_stride, p25fe_result_t *d_result,
                    void *stream);

/* stages 1-7 from a fresh stream state over a resident capture (BASELINE.json config 2/4). */
int p25fe_run_dev(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n, uint8_t *d_dibits,
                  size_t dibit_stride, p25fe_result_t *d_result, void *stream);

/* The same pass with the two halves of the path on two streams, for a SEQUENCE of captures (or chunks of separate
 * tuners): stages 1-5 (K1, which is what loads the GPU) run on `stream`, stages 6-7 (sync detect, scan, slicer: three
 * short, latency-bound launches) on a stream of the handle, so that they overlap the NEXT call's K1 -- the way
 * DemodTask and RecvTask are two threads joined by a channel (src/demod.rs:116, src/recv.rs:140-150: the demodulator
 * is already filling the next chunk while the receiver works on the previous one).  The receiver's scratch is double
 * buffered; call i + 2 waits on the device for call i's receive kernels.
 * d_dibits / d_result of a call are complete only after p25fe_join_dev has been enqueued on a stream and that stream
 * has reached it (or after any other call on this handle that uses the receiver, which joins first).  Consecutive calls
 * may name the same output buffers (their receive kernels run in call order on one stream); d_iq must stay unchanged
 * until `stream` has passed the call (K1 is the only reader). */
int p25fe_run_dev_pipelined(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n, uint8_t *d_dibits,
                            size_t dibit_stride, p25fe_result_t *d_result, void *stream);
/* make `stream` wait for every receive kernel that p25fe_run_dev_pipelined has enqueued so far */
int p25fe_join_dev(p25fe_t *h, void *stream);

/* Time-sharded capture (BASELINE.json config 5): shard = owned range [abs0, abs0 + n) with
 * n_hist >= p25fe_shard_halo() samples of left context in memory (or n_hist == abs0 for the
 * first shard).  Pass 1 demodulates, detects frame syncs and fills d_result[c] with the
 * shard summary (first_event, n_dibits_after_first, anchor_out) assuming no carry-in.  The
 * caller exchanges summaries (all_gather of a few bytes), resolves each shard's carry-in
 * anchor with p25fe_shard_resolve on the host, then pass 2 slices.
 * ORDERING: pass 2 works on the baseband and the detections that pass 1 left in the handle's
 * scratch.  No other call that demodulates or slices (p25fe_run_*, p25fe_demod_*, p25fe_slice*,
 * another shard's pass 1) may come between a shard's pass 1 and its pass 2 on the same handle:
 * any of them invalidates the shard context and pass 2 then returns P25FE_ERR_ARG (as it does
 * after a failed pass 1) instead of slicing stale data. */
size_t p25fe_shard_halo(void);
int p25fe_shard_pass1(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                      uint64_t abs0, p25fe_result_t *d_result, void *stream);
int p25fe_shard_pass2(p25fe_t *h, const p25fe_anchor_t *d_anchor_in, uint8_t *d_dibits, size_t dibit_stride,
                      p25fe_result_t *d_result, void *stream);
/* The same pass 1 in two launches, so that the exchange of the filter-state overlap hides behind K1: _main enqueues
 * the front end for every part of the shard whose input lies inside the OWNED samples (all but the first ~900
 * baseband samples) and may run while the left neighbour's halo is still on the wire; _finish (same arguments, after
 * the halo has arrived in front of d_iq) enqueues the remaining head, the sync detection and the scan, and fills
 * d_result.  p25fe_shard_pass1 == _main + _finish. */
int p25fe_shard_pass1_main(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                           uint64_t abs0, void *stream);
int p25fe_shard_pass1_finish(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                             uint64_t abs0, p25fe_result_t *d_result, void *stream);
/* The head alone (same arguments, after the halo has arrived), so that it can run on ANOTHER stream beside the main launch
 * -- the stream the halo arrives on: the two launches share no byte of their outputs.  p25fe_shard_pass1_finish then only
 * enqueues the sync detection and the scan, and needs NO stream-level wait for the head: the head launch's last workgroup
 * publishes a flag word, and the detection workgroups whose tiles read the head's planes poll it (a cross-stream event wait
 * costs 10 - 20 us of idle GPU) -- for at most 2 s of device wall clock, after which they give up and p25fe_shard_head_check
 * reports P25FE_ERR_TIMEOUT (a stalled peer must not hang the queue).  Making _finish's stream wait for the head's with an
 * event as well is harmless.  Optional: without this call _finish launches the head itself. */
int p25fe_shard_pass1_head(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                           uint64_t abs0, void *stream);
/* After the stream(s) of a step whose head ran through p25fe_shard_pass1_head have been synchronised: P25FE_OK, or P25FE_ERR_TIMEOUT if a
 * detection workgroup gave up waiting for the head's flag (2 s of device wall clock: the stream the head was on never delivered it, e.g. a
 * halo exchange whose peer died) -- the results of that step, and of every later one on this handle, are not to be used. */
int p25fe_shard_head_check(p25fe_t *h);
/* The whole front end of pass 1 in ONE launch (main + head: the halo must be in memory), nothing else; p25fe_shard_pass1_finish
 * then only enqueues the sync detection and the scan.  p25fe_shard_pass1 == _k1 + _finish. */
int p25fe_shard_pass1_k1(p25fe_t *h, const void *d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                         void *stream);
/* Steps one behind the other (p25fe_shard_step_pipelined): between _begin and _end the shard passes work on the next of up to four
 * scratch sets (p25fe_run_dev_pipelined rotates through two of them) and do not wait for the handle's receive stream -- they are
 * its work.  _begin makes `stream` (where pass 1's front end is about to overwrite that set's planes) wait for the passes that read
 * it four steps back and returns the receive stream in *rx_stream; p25fe_shard_pass1_k1 / _main(..., stream) then makes the receive
 * stream wait for its launch; every later pass of the step (_finish, pass 2, the exchanges between them) is enqueued on the receive
 * stream or on streams that wait for it.  _end(h, last) records "this step's passes are done" on `last` (NULL: the receive stream)
 * and marks it as pending: p25fe_join_dev makes a stream wait for it, every non-pipelined entry point does so by itself. */
int p25fe_shard_pipe_begin(p25fe_t *h, void *stream, void **rx_stream);
int p25fe_shard_pipe_end(p25fe_t *h, void *last_stream);
/* the handle's receive stream (created on first use), without starting a step: for a host that wants to check its own streams against it
 * (p25fe_streams_share_queue; libp25fe_rccl.so's p25fe_shard_create / p25fe_shard_prepare do) */
int p25fe_rx_stream(p25fe_t *h, void **rx_stream);

/* Host-side combine: summaries[r] for r = 0..n_shards-1 in time order (one channel) ->
 * anchor_in[r] (n_shards entries) and dibit_offset[r] (n_shards + 1 entries: shard r's dibits occupy
 * [dibit_offset[r], dibit_offset[r + 1]) of the whole stream, the last entry is the total).
 * Pure integer/struct logic, no device work.  symbol_clock: the handles' p25fe_config_t.symbol_clock. */
int p25fe_shard_resolve(const p25fe_result_t *summaries, const uint64_t *shard_bb0, const uint64_t *shard_bb_n,
                        size_t n_shards, int symbol_clock, p25fe_anchor_t *anchor_in, uint64_t *dibit_offset);

/* Same combine on the device (one tiny kernel, no host synchronisation): all arrays are device pointers, e.g. the
 * all-gathered summaries; writes n_shards anchors and n_shards + 1 offsets; pass d_anchor_in + rank to p25fe_shard_pass2. */
int p25fe_shard_resolve_dev(p25fe_t *h, const p25fe_result_t *d_summaries, const uint64_t *d_shard_bb0,
                            const uint64_t *d_shard_bb_n, size_t n_shards, p25fe_anchor_t *d_anchor_in,
                            uint64_t *d_dibit_offset, void *stream);

/* Pass 2 with the combine on the device and INSIDE the pass: d_summaries = the n_shards pass-1 summaries in time order
 * (the all-gathered records, device memory), `rank` = this handle's shard.  One-channel handles.  With the fixed-stride
 * receiver and no lock drops inside the shard this is ONE launch: no second scan and no resolve kernel -- every slicer
 * workgroup derives its carry-in from the summaries and applies it in closed form (ShardFix, p25fe_recv.hip); otherwise it
 * enqueues p25fe_shard_resolve_dev + p25fe_shard_pass2.  Also written: d_anchor_in[n_shards] and
 * d_dibit_offset[n_shards + 1] (what p25fe_shard_resolve_dev writes), *d_result = this shard's final record.
 * d_dibits_dup (nullable): a second destination of every dibit with the same row layout -- rank 0's shard starts at offset
 * 0 of the capture's ordered stream, so it can slice straight into it.
 * Same ordering rule as p25fe_shard_pass2: it follows its shard's pass 1 on this handle. */
int p25fe_shard_pass2_dev(p25fe_t *h, const p25fe_result_t *d_summaries, const uint64_t *d_shard_bb0,
                          const uint64_t *d_shard_bb_n, size_t n_shards, size_t rank, p25fe_anchor_t *d_anchor_in,
                          uint64_t *d_dibit_offset, uint8_t *d_dibits, size_t dibit_stride, uint8_t *d_dibits_dup,
                          p25fe_result_t *d_result, void *stream);

/* Dibit gather, receiving side (BASELINE.json config 5: "RCCL ... carrying ... the reduced dibit stream"): after an
 * all-gather of the shards' dibit buffers (each `cap` bytes, shard r at d_gathered + r * cap, its first
 * dibit_offset[r + 1] - dibit_offset[r] bytes valid) write the contiguous stream to d_out[0 .. dibit_offset[n_shards]).
 * d_dibit_offset: the n_shards + 1 entries of p25fe_shard_resolve_dev.  The consumer is the ordered stream that
 * RecvTask feeds into MessageReceiver (src/recv.rs:148-150). */
int p25fe_shard_compact_dev(p25fe_t *h, const uint8_t *d_gathered, size_t cap, const uint64_t *d_dibit_offset,
                            size_t n_shards, uint8_t *d_out, size_t out_cap, void *stream);
/* the same for shards first_shard .. n_shards - 1 only (first_shard = 1: rank 0 has sliced its own shard into d_out already,
 * p25fe_shard_pass2_dev's d_dibits_dup; first_shard == n_shards: nothing to do) */
int p25fe_shard_compact_from_dev(p25fe_t *h, const uint8_t *d_gathered, size_t cap, const uint64_t *d_dibit_offset,
                                 size_t first_shard, size_t n_shards, uint8_t *d_out, size_t out_cap, void *stream);

/* Do two streams of this process share a HARDWARE queue (*shared = 1) -- i.e. do their kernels run one after the other whatever the
 * streams say?  HIP multiplexes the streams of one priority level onto four queues in creation order; the pipelined calls of this
 * library only overlap what they promise when their streams sit on different queues ("Which stream to pass").
 * Synchronises both streams and runs a 1-thread probe on each (< 1 ms); p25fe_shard_create's side stream is chosen with it. */
int p25fe_streams_share_queue(p25fe_t *h, void *stream_a, void *stream_b, int *shared);

/* Network identifier that follows each frame sync (next row after the dibits, SURVEY.md section 8f: what
 * p25::MessageReceiver reports as MessageEvent::PacketNID, src/recv.rs:216-222, consumed by
 * ReceiverPolicy::handle_nid, src/policy.rs:92).  64 bits = BCH(63,16,23) code word (NAC 12 bits, DUID 4 bits)
 * + 1 extra bit, 32 dibits after the sync word with a status symbol interleaved (docs/SPEC.md 3.9). */
typedef struct p25fe_nid {
    uint64_t raw;                        /* received 64 bits, first transmitted bit in bit 63 */
    int64_t sync_pos;                    /* copied from the sync event (0 if d_sync_pos is null) */
    uint16_t nac;                        /* network access code of the nearest code word */
    uint8_t duid;                        /* data unit id */
    uint8_t n_errors;                    /* Hamming distance to that code word */
    int32_t valid;                       /* 1 decoded (<= 11 bit errors), 0 undecodable, -1 dibit stream ends inside the NID */
} p25fe_nid_t;

/* d_dibits / n_dibits: one channel's dibit stream (as written by p25fe_run_dev / p25fe_slice_dev); d_sync_dibit[k]:
 * index of the first dibit after sync word k (the sync_dibit output of p25fe_slice_dev); d_sync_pos nullable.
 * Writes n_sync records.  All device pointers; enqueued on `stream`. */
int p25fe_nid_dev(p25fe_t *h, const uint8_t *d_dibits, size_t n_dibits, const uint64_t *d_sync_dibit,
                  const int64_t *d_sync_pos, size_t n_sync, p25fe_nid_t *d_out, void *stream);

/* Host-buffer form (the file-driven harness: p25fe_replay -j): copies the stream and the events to the device, runs the
 * same kernel, copies the n_sync records back; synchronous. */
int p25fe_nid(p25fe_t *h, const uint8_t *dibits, size_t n_dibits, const uint64_t *sync_dibit, const int64_t *sync_pos,
              size_t n_sync, p25fe_nid_t *out);

/* The same for a whole channel batch without a host round trip: channel c's stream is d_dibits + c * dibit_stride
 * with d_result[c].n_dibits dibits, its events d_sync_dibit / d_sync_pos + c * sync_stride (the outputs of
 * p25fe_slice_dev), min(d_result[c].n_sync, sync_stride) of them; records go to d_out + c * sync_stride. */
int p25fe_nid_batch_dev(p25fe_t *h, const uint8_t *d_dibits, size_t dibit_stride, const p25fe_result_t *d_result,
                        const uint64_t *d_sync_dibit, const int64_t *d_sync_pos, size_t sync_stride,
                        p25fe_nid_t *d_out, void *stream);

/* Polyphase channeliser (no reference counterpart -- the reference tunes ONE channel,
 * src/sdr.rs:64-65): one cf32 capture at 2.4 Msps -> P25FE_CHZ_CHANNELS = 192 channels on the 12.5 kHz raster
 * (channel c centred c * 12.5 kHz above the capture's centre, c >= 96 below it), each at 240 ksps -- the stream
 * stage 1 (src/demod.rs:74-84) would deliver for a tuner on that channel, ready for p25fe_demod_dev / p25fe_run_dev
 * of a 192-channel handle (ch_stride = out_stride).  docs/SPEC.md 3.11; channel 0 equals p25fe_predecim_dev.
 * d_iq: owned sample 0 (16-byte aligned), n_hist valid samples before it (>= 79 for exact continuation), abs0 its
 * absolute index (fixes the decimation grid and the mixer phase, so any chunking gives the same stream);
 * d_out: [192][out_stride] cf32, n_out = p25fe_n_predecim(abs0, n) samples per channel; rows are written in whole
 * 64-sample tiles, so out_stride >= n_out rounded up to 64 (P25FE_ERR_ARG otherwise) and row samples >= n_out are
 * unspecified.  The handle only supplies the device. */
#define P25FE_CHZ_CHANNELS_ABI 192
int p25fe_channelise_dev(p25fe_t *h, const float *d_iq, size_t n_hist, size_t n, uint64_t abs0, float *d_out,
                         size_t out_stride, void *stream);

/* Per-channel observability record: what the reference pushes to its hub as
 * HubEvent::UpdateSignalPower (src/demod.rs:95-101, "sigPower" src/hub.rs:344) and HubEvent::UpdateStats
 * (src/recv.rs:162-165, 212-215; "updateStats" src/hub.rs:401-402), restricted to what exists on this path: the
 * "bch" row of serialize_stats (src/hub.rs:559) -- p25::stats::CodeStats as serialised by serialize_code_stats
 * (src/hub.rs:574-581: totalWords = words, errWords = errs, totalSymbols = words * size, fixedSymbols = fixed) --
 * plus the receiver's lock state.  The other code rows (Golay, Hamming, RS, Viterbi) belong to the out-of-scope
 * frame decoder. */
typedef struct p25fe_code_stats {
    uint64_t words;                      /* code words seen (NIDs complete in the dibit stream) */
    uint64_t errs;                       /* of those, undecodable (more than 11 bit errors) */
    uint64_t fixed;                      /* corrected bits over the decodable words */
    uint32_t size;                       /* symbols per word: 63 */
    uint32_t reserved;
} p25fe_code_stats_t;

typedef struct p25fe_chan_stats {
    float sig_power_dbm;                 /* copied from d_power_dbm[c]; NaN when not supplied */
    int32_t locked;                      /* an anchor is in force at the end of the range */
    uint64_t n_dibits;
    uint64_t n_sync;                     /* detections decided in the range */
    int64_t last_sync_pos;               /* s of the last detection, -1 if none */
    p25fe_code_stats_t bch;
} p25fe_chan_stats_t;

/* d_result[C] from p25fe_slice_dev / p25fe_run_dev; d_nid[C][sync_stride] from p25fe_nid_batch_dev (nullable: bch
 * stays zero); d_power_dbm[C] from p25fe_demod_dev (nullable).  Writes d_stats[C]. */
int p25fe_chan_stats_dev(p25fe_t *h, const p25fe_result_t *d_result, const p25fe_nid_t *d_nid, size_t sync_stride,
                         const float *d_power_dbm, p25fe_chan_stats_t *d_stats, void *stream);

/* Measurement hook for bench.py: when enabled, p25fe_run_dev / p25fe_shard_pass1 record HIP
 * events on the caller's stream around each kernel (K1 front end, K2 sync detect, K3 scan, K4 slice);
 * p25fe_shard_pass2 records its own slot with K3 and K4 only (ms[0], ms[1] of that slot read as 0).
 * p25fe_profile_read synchronises those events and returns the summed milliseconds per kernel over the
 * kept slots (at most the last 64) and, in *n_calls, how many of those slots ran K1 (= passes of the path).
 * on = 1: events around every kernel (five records per call); on = 2: around K1 only (two records; ms[1..3] read
 * as 0) -- the records themselves cost ~4 us each between kernels, which matters for a 0.28 ms step; on = 3: as 2, but
 * only every 8th call carries the two records (the kept slots then span the last 512 calls). */
int p25fe_profile_enable(p25fe_t *h, int on);
int p25fe_profile_read(p25fe_t *h, double ms[4], uint64_t *n_calls);

/* Number of baseband samples produced by n input samples starting at absolute index abs0 -- with the default decimator
 * phase (p25fe_config_t.decim_phase = 4), and with the handle's. */
size_t p25fe_n_baseband(uint64_t abs0, size_t n);
size_t p25fe_n_baseband_h(const p25fe_t *h, uint64_t abs0, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* P25FE_H */
