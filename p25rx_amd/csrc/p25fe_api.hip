// p25fe_api.hip -- host side of the C ABI declared in include/p25fe.h.
//
// Owns device scratch and the stream state that DemodTask / MessageReceiver keep in their
// structs (src/demod.rs:25-40, src/recv.rs:47), and launches the kernels of p25fe_kernels.hip.
// No CPU compute path exists here: every entry point either runs the HIP kernels or fails.
#include "p25fe_kernels.hip"

#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "p25fe_jit.h"

using namespace p25k;

namespace {

constexpr size_t HISTPAD = 704;      // >= HIST_IQ_MAX (702 with 64 + 64 + 64 taps), multiple of 8: keeps 16-B alignment for u8 and cf32
static_assert(HISTPAD >= (size_t)HIST_IQ_MAX && HISTPAD % 8 == 0, "stream history covers the longest filters");
constexpr size_t BBPAD = 256;        // >= HIST_BB (240) + the tracking clock's lookahead, multiple of 4
constexpr size_t SHARD_HALO = 2560;                     // samples (20 KB of cf32): K1 recomputes one block (320 + 2) of baseband history in front of a
                                                       // shard and needs up to 127 decimator outputs (64-tap post-discriminator and channel filters;
                                                       // 160 when a segment recomputes its halo) and 63 samples in front of that
static_assert(SHARD_HALO >= (size_t)(DEC * (PLPAD + CLK_L + 160) + TMAX) && 160 >= TMAX + TMAX - 1 && SHARD_HALO % 8 == 0, "shard halo");
constexpr uint32_t STATE_MAGIC = 0x50323546u;          // "P25F"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// pinned host memory that the device reads and writes in place (the streaming entry points' staging and result buffers)
struct PinBuf {
    void* p = nullptr;          // host view
    void* dp = nullptr;         // device view of the same bytes
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        // coherent (fine-grained): the chunk kernels' results and completion word must reach the polling host while the
        // kernel is still running, not at its end
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostMalloc(&p, want, hipHostMallocDefault); }
        if (e != hipSuccess) { p = nullptr; return e; }
        e = hipHostGetDevicePointer(&dp, p, 0);
        if (e != hipSuccess) { (void)hipHostFree(p); p = dp = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = dp = nullptr; cap = 0; }
};

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// carve `bytes` (rounded up to 64) out of a pinned arena; returns the host pointer, *dev = the device view
struct Arena {
    char* hp; char* dp; size_t off = 0;
    Arena(const PinBuf& b) : hp(static_cast<char*>(b.p)), dp(static_cast<char*>(b.dp)) {}
    template <class T> T* take(size_t count, T** dev)
    {
        T* r = reinterpret_cast<T*>(hp + off);
        *dev = reinterpret_cast<T*>(dp + off);
        off += round_up(count * sizeof(T), 64);
        return r;
    }
};



// Kernel launch with optional events ATTACHED TO THE DISPATCH (hipExtLaunchKernelGGL): the start / stop events are
// signalled by the kernel's own AQL packet, where hipEventRecord would put one more barrier packet (and its release) in
// front of the next kernel of the stream.
template <class... Args, class... Act>
inline void launch_ev(void (*kern)(Args...), dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t e0, hipEvent_t e1,
                      Act... args)
{
    if (e0 || e1) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, st, e0, e1, 0u, Args(args)...);
    else hipLaunchKernelGGL(kern, grid, block, lds, st, Args(args)...);
}
inline size_t fmt_bytes(int fmt) { return fmt == P25FE_FMT_CF32 ? 8 : 2; }

}  // namespace

static_assert(BBPAD == (size_t)TAILN && BBPAD >= (size_t)HIST_BB + CLK_L, "baseband tail kept between calls");

constexpr int MAX_LANES = 4;
// p25fe::sh_flag: words 0 / 1 the head segment's flag and ticket, 2 / 3 the queue probe's pair, 4 "a detection gave up waiting for the head"
constexpr int SH_FLAG_WORDS = 8, SH_FLAG_ERR = 4;

struct p25fe {
    p25fe_config_t cfg;
    int C = 1;
    int last_hip = 0;
    int n_cu = 256;
    Taps taps;                             // resolved numbers: padded tables, the u8 table, the discriminator's scale
    int k1_p = 5;                          // FIR outputs per thread
    int variant = P25FE_VARIANT_BUILTIN;   // which front-end kernels run (p25fe_kernel_variant)
    bool long_taps = false;                // more than P25FE_T1 / P25FE_T2 taps -> the 64 / 64 geometry (Geo<5, 1>)
    int phase = P25FE_DECIM_PHASE;         // p25fe_config_t.decim_phase: baseband sample m comes from input 5 m + phase
    int n_avg = BOX;                       // post-discriminator filter: taps in use (the table is taps.avg)
    bool u8_lut_mode = false;              // the u8 table is not affine: the specialised kernels look it up in LDS (the generic ones always do)
    DevBuf d_taps;                         // device copy (generic kernels: everything; specialised ones: the u8 table, if not affine)
    hipModule_t jit_mod = nullptr;         // specialised kernels (p25fe_jit.cpp): [format][linear, planar, chunk]
    hipFunction_t jit_fn[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipStream_t stream = nullptr;          // for the host-pointer calls

    int track = 0;                         // p25fe_config_t.symbol_clock without its flag bits (docs/SPEC.md 3.8b)
    bool causal_ok = false;                // P25FE_CLOCK_CAUSAL_OK: mode 2 may run as mode 1 in the calls that see the stream in pieces
    long look = 0;                         // samples the receiver runs behind the baseband (2 with the tracking clock)
    // lock drops for the next receiver-running call (p25fe_resync_at_dev); consumed by it
    const long* rs_idx = nullptr;
    size_t rs_n = 0, rs_stride = 0;
    // scratch
    DevBuf pl_f, pl_bits, evl, evthr, recs, tsum, outs, power_partial, chunk_cnt;
    DevBuf gagg, gpre, gtick;              // K3 (k_scan_tiles): group aggregates, group carry-ins, the arrival counters (zero between launches)
    PinBuf hin, hbb, hout;                 // streaming entry points: staged input ([history | new] IQ, [tail | new] baseband), results
    DevBuf gsum, gouts, evg, gsg, gpg;     // general receiver only (tracking clock / lock drops): allocated on first use
    DevBuf evrec, evnext, evoff;           // SPEC 3.8c (symbol_clock = 2, resident ranges): the list of detections, allocated on first use
    unsigned long long ev_seq = 0;         // ... and the sequence number that marks an EvNext entry as this call's
    // p25fe_run_dev_pipelined: a second set of the receiver's scratch (the member names above always are the set of the
    // current call; the sets are swapped per call), the stream the receive kernels run on, and the events that order
    // K1 (caller's stream) -> K2..K4 (rx_stream) -> next K1 into the same set two calls later
    // (p25fe_run_dev_pipelined rotates through two sets; the pipelined shard step through up to MAX_LANES: its chain behind K1
    // holds exchanges that only find room on the chip when a K1 launch drains, so it spans more than one K1.)  A set keeps its id
    // (`lane` = the current one's); the events and the pending flags are indexed by it.
    struct RxSet { DevBuf pl_f, pl_bits, evl, evthr, recs, tsum, outs, gsum, gouts, evg, gagg, gpre, gtick, gsg, gpg; int id = 0; };
    RxSet spare[MAX_LANES - 1];
    hipStream_t rx_stream = nullptr;
    hipEvent_t ev_k1[MAX_LANES] = {}, ev_rx[MAX_LANES] = {};
    bool rx_pending[MAX_LANES] = {};
    hipStream_t rx_joined[MAX_LANES] = {};           // stream that has already been made to wait for ev_rx[l] (valid while rx_pending[l])
    bool rx_joined_any[MAX_LANES] = {};
    int lane = 0;
    int sh_depth = MAX_LANES;              // sets the pipelined shard step rotates through (P25FE_SHARD_PIPE_DEPTH)
    int run_depth = 2;                     // sets p25fe_run_dev_pipelined rotates through (P25FE_PIPE_DEPTH)
    // p25fe_run_host_windows: two device windows, dibit rows and result records in a ring, the copy streams and their events
    DevBuf win_buf[2], win_dib[2], win_res, win_anc;
    PinBuf win_stage[2], win_out;
    hipStream_t win_cs = nullptr, win_os = nullptr;
    hipEvent_t win_ev[4][5] = {};          // per ring slot: H2D begin / end, compute begin / end, results on the host
    bool ext_events = true;                // events ride on the kernel dispatches (hipExtLaunchKernelGGL) instead of separate records
    int rx_cus = 0;                        // > 0: the receive stream is confined to this many CUs (hipExtStreamCreateWithCUMask)
    // stream state (per channel, channel-major in the device buffers)
    unsigned chunk_seq = 0;                // completion sequence number of the one-launch chunk calls
    uint64_t abs_iq = 0;                   // IQ samples consumed
    int fmt_locked = -1;
    std::vector<char> hist_iq;             // [C][SHARD_HALO] raw samples (8 bytes reserved per sample), host side
    uint64_t abs_bb = 0;                   // baseband samples consumed by the slicer
    std::vector<float> tail_bb;            // [C][BBPAD], host side
    std::vector<p25fe_anchor_t> anchor;    // [C]
    std::vector<uint64_t> total_dibits;    // [C]
    // profiling ring: PROF_RING calls x 5 events (before K1, after K1, K2, K3, K4)
    bool prof_on = false;
    int prof_level = 1;             // 1: events around every kernel, 2: around K1 only (two records per call, not five)
    std::vector<hipEvent_t> prof_ev;
    std::vector<uint8_t> prof_mask;        // per slot: which of its five events were recorded
    uint64_t prof_calls = 0;
    uint64_t prof_seq = 0;                 // level 3: calls seen (every PROF_SAMPLE-th is recorded)
    int prof_slot = -1;                    // slot being recorded by the current call
    // shard context between pass1 and pass2 (the planar baseband and the tile summaries stay in the scratch buffers;
    // every other entry point that touches them clears sh_nbb, so a stale pass 2 fails instead of slicing garbage)
    bool sh_valid = false;
    size_t sh_main_nbb = (size_t)-1;       // shard whose main K1 launch is out, waiting for p25fe_shard_pass1_finish
    uint64_t sh_main_abs0 = 0;
    bool sh_head_done = false;             // ... and whose head segment has been launched too (p25fe_shard_pass1_head)
    bool sh_head_flagged = false;          // ... by p25fe_shard_pass1_head: the detection's first tiles wait for sh_flag == sh_seq
    DevBuf sh_flag;                        // one word a one-thread kernel behind the head writes
    unsigned sh_seq = 0;
    int sh_head_tile_max = -1;
    size_t sh_nbb = 0;
    long sh_abs_bb0 = 0;
    bool sh_gen = false;                   // pass 1 ran the general receiver (pass 2 reads its summaries)
    bool sh_pipe = false;                  // between p25fe_shard_pipe_begin and _end: the shard passes do not join the receive stream (they ARE its work)
    bool sh_scan_fresh = false;            // the groups' carry-ins (`gpre`) are still pass 1's (no carry-in): p25fe_shard_pass2 rewrites them
};

// geometry of the planar scratch for n_bb owned baseband samples (p25fe_recv.hip: Planar)
struct PlanarGeo {
    size_t n_tiles, n_blocks;
    explicit PlanarGeo(size_t n_bb)
    {
        n_tiles = (n_bb + TS - 1) / TS;
        if (n_tiles == 0) n_tiles = 1;
        n_blocks = (size_t)TWORDS * n_tiles + 6;   // K1 writes symbols < 768 n_tiles + 32; K2 reads words < 24 n_tiles + 5
    }
    size_t floats() const { return n_blocks * PL_BLK; }
    size_t words() const { return n_blocks * SPS; }
};

// Symbol indices are 32-bit in the receive kernels: a range holds < 2^31 symbols (124 h of one channel).  Checked by every
// entry point BEFORE it sizes scratch for the range, so that an absurd length is an argument error, not an allocation failure.
constexpr size_t MAX_RANGE_BB = (size_t)0x7ff00000u * 10u;

constexpr int PROF_RING = 64;
constexpr int PROF_SAMPLE = 8;

// --------------------------------------------------------------------------------------------
// p25fe_config_t -> the numbers the kernels run with.  No device needed (p25fe_specialize runs on a build host).
// --------------------------------------------------------------------------------------------
struct Resolved {
    Taps taps;              // tables zero-padded to the evaluation length, the u8 table, the discriminator's scale, the post-discriminator filter
    bool long_taps;         // 64 / 64 evaluation
    bool lut_affine;        // the u8 table is fma(b, u8_scale, u8_offset) for every byte
    float u8_scale, u8_offset;
    bool dflt;              // every number is the build's own (p25fe_spec.h): the built-in immediate-coefficient kernels apply
};

static inline bool finite_f(float v) { return v - v == 0.0f; }
static inline bool same_bits(float a, float b) { return memcmp(&a, &b, sizeof a) == 0; }

static bool lut_is_affine(const float* lut, float sc, float of)
{
    for (int b = 0; b < 256; ++b)
        if (!same_bits(lut[b], fmaf((float)b, sc, of))) return false;
    return true;
}

static int resolve_config(const p25fe_config_t* cfg, Resolved* r)
{
    if (cfg->abi_version != P25FE_ABI_VERSION || cfg->n_decim_taps < 1 || cfg->n_decim_taps > P25FE_MAX_TAPS || cfg->n_chan_taps < 1 ||
        cfg->n_chan_taps > P25FE_MAX_TAPS || (cfg->symbol_clock != P25FE_CLOCK_FIXED && cfg->symbol_clock != P25FE_CLOCK_TRACKING && cfg->symbol_clock != P25FE_CLOCK_TRACKING_RESLICE &&
         cfg->symbol_clock != (P25FE_CLOCK_TRACKING_RESLICE | P25FE_CLOCK_CAUSAL_OK)) ||
        cfg->specialize < P25FE_SPECIALIZE_OFF || cfg->specialize > P25FE_SPECIALIZE_FORCE || cfg->decim_phase < 0 || cfg->decim_phase >= DEC ||
        cfg->n_avg_taps < 1 || cfg->n_avg_taps > P25FE_MAX_TAPS)
        return P25FE_ERR_ARG;
    memset(&r->taps, 0, sizeof r->taps);                             // zero padding at the old end is bit-neutral on finite samples
    // MovingAverage::new(10), src/demod.rs:52 (docs/SPEC.md 3.5): all taps equal -> a moving average, summed then scaled once
    r->taps.n_avg = cfg->n_avg_taps;
    r->taps.avg_uniform = 1;
    for (int k = 0; k < cfg->n_avg_taps; ++k) {
        if (!finite_f(cfg->avg_taps[k])) return P25FE_ERR_ARG;
        r->taps.avg[k] = cfg->avg_taps[k];
        if (!same_bits(cfg->avg_taps[k], cfg->avg_taps[0])) r->taps.avg_uniform = 0;
    }
    for (int k = 0; k < cfg->n_decim_taps; ++k) { if (!finite_f(cfg->decim_taps[k])) return P25FE_ERR_ARG; r->taps.dec[k] = cfg->decim_taps[k]; }
    for (int k = 0; k < cfg->n_chan_taps; ++k) { if (!finite_f(cfg->chan_taps[k])) return P25FE_ERR_ARG; r->taps.ch[k] = cfg->chan_taps[k]; }
    r->long_taps = cfg->n_decim_taps > P25FE_T1 || cfg->n_chan_taps > P25FE_T2;
    // FmDemod::new(deviation, sample_rate), src/demod.rs:54 -> output scale (docs/SPEC.md 3.4)
    if (cfg->fm_gain != 0.0f) {
        if (!finite_f(cfg->fm_gain)) return P25FE_ERR_ARG;
        r->taps.fm_gain = cfg->fm_gain;
    } else {
        if (cfg->fm_deviation_hz == 0 || cfg->fm_sample_rate_hz == 0) return P25FE_ERR_ARG;
        r->taps.fm_gain = (float)((double)cfg->fm_sample_rate_hz / (2.0 * M_PI * (double)cfg->fm_deviation_hz));
    }
    // rtlsdr_iq::IQ, src/demod.rs:83 (docs/SPEC.md 3.1)
    if (cfg->u8_lut_valid) {
        for (int b = 0; b < 256; ++b) { if (!finite_f(cfg->u8_lut[b])) return P25FE_ERR_ARG; r->taps.lut[b] = cfg->u8_lut[b]; }
        // a table that IS an fma of the byte runs as arithmetic: the caller's own pair first, then the two obvious fits
        const float* L = r->taps.lut;
        const float cand[3][2] = {{cfg->u8_scale, cfg->u8_offset}, {L[1] - L[0], L[0]},
                                  {(float)(((double)L[255] - (double)L[0]) / 255.0), L[0]}};
        r->lut_affine = false;
        for (int c = 0; c < 3 && !r->lut_affine; ++c)
            if (finite_f(cand[c][0]) && finite_f(cand[c][1]) && lut_is_affine(L, cand[c][0], cand[c][1])) {
                r->lut_affine = true; r->u8_scale = cand[c][0]; r->u8_offset = cand[c][1];
            }
        if (!r->lut_affine) r->u8_scale = r->u8_offset = 0.0f;
    } else {
        if (!finite_f(cfg->u8_scale) || !finite_f(cfg->u8_offset)) return P25FE_ERR_ARG;
        r->lut_affine = true; r->u8_scale = cfg->u8_scale; r->u8_offset = cfg->u8_offset;
        for (int b = 0; b < 256; ++b) r->taps.lut[b] = fmaf((float)b, cfg->u8_scale, cfg->u8_offset);
    }
    Taps def;
    memset(&def, 0, sizeof def);
    memcpy(def.dec, P25FE_DEFAULT_DECIM_TAPS, sizeof(float) * P25FE_T1);
    memcpy(def.ch, P25FE_DEFAULT_CHAN_TAPS, sizeof(float) * P25FE_T2);
    memcpy(def.avg, P25FE_DEFAULT_AVG_TAPS, sizeof(float) * BOX);
    r->dflt = !r->long_taps && memcmp(def.dec, r->taps.dec, sizeof def.dec) == 0 && memcmp(def.ch, r->taps.ch, sizeof def.ch) == 0 &&
              r->taps.n_avg == BOX && memcmp(def.avg, r->taps.avg, sizeof def.avg) == 0 &&
              same_bits(r->taps.fm_gain, P25FE_FM_GAIN) && r->lut_affine && same_bits(r->u8_scale, P25FE_U8_SCALE) &&
              same_bits(r->u8_offset, P25FE_U8_OFFSET);
    return P25FE_OK;
}

static p25jit::Spec jit_spec(const Resolved& r)
{
    p25jit::Spec s;
    memset(&s, 0, sizeof s);
    s.tx = r.long_taps ? 1 : 0;
    s.t1 = r.long_taps ? TMAX : T1;
    s.t2 = r.long_taps ? TMAX : T2;
    memcpy(s.dec, r.taps.dec, sizeof s.dec);
    memcpy(s.ch, r.taps.ch, sizeof s.ch);
    s.fm_gain = r.taps.fm_gain;
    s.u8_lut = r.lut_affine ? 0 : 1;
    s.u8_scale = r.u8_scale; s.u8_offset = r.u8_offset;
    s.n_avg = r.taps.n_avg; s.avg_uniform = r.taps.avg_uniform;
    memcpy(s.avg, r.taps.avg, sizeof s.avg);
    return s;
}

static thread_local std::string t_jit_log;          // compiler log of this thread's last p25fe_create / p25fe_specialize

// Where specialised code objects are looked for: $P25FE_SPEC_DIR (a deployment's ahead-of-time directory), then the cache.
static std::vector<std::string> jit_dirs()
{
    std::vector<std::string> d;
    const char* e = getenv("P25FE_SPEC_DIR");
    if (e && *e) d.push_back(e);
    d.push_back(p25jit::default_cache_dir());
    return d;
}

// P25FE_SPECIALIZE_AUTO fell back to the generic kernels (1.3 - 2.4 x slower): say so once per process, on stderr -- a caller
// that never polls p25fe_kernel_variant would otherwise not know.  P25FE_QUIET=1 silences it.
static void fallback_notice()
{
    static bool said = false;
    if (said) return;
    said = true;
    const char* q = getenv("P25FE_QUIET");
    if (q && atoi(q) != 0) return;
    fprintf(stderr, "p25fe: no specialised kernels for this configuration's numbers (no usable cached code object and hipRTC did not "
                    "deliver one): running the GENERIC kernels, 1.3 - 2.4 x slower.  p25fe_specialize_log() has the details; "
                    "p25fe_specialize() / $P25FE_SPEC_DIR is the ahead-of-time form.\n");
}

// Which front-end kernels a handle made from cfg gets on this host, and -- for P25FE_VARIANT_SPECIALIZED -- the code object
// (looked up in $P25FE_SPEC_DIR and the cache, else compiled and stored).  Needs no device.  Returns the variant or P25FE_ERR_JIT.
static int choose_variant(const p25fe_config_t* cfg, const Resolved& rs, std::vector<char>* code, std::string* path, bool* from_cache)
{
    // P25FE_JIT=0 in the environment switches the AUTO mode off for a whole process.
    static const bool jit_env_off = [] { const char* v = getenv("P25FE_JIT"); return v && atoi(v) == 0; }();
    const bool must = cfg->specialize >= P25FE_SPECIALIZE_REQUIRE;
    const bool want_jit = (!rs.dflt || cfg->specialize == P25FE_SPECIALIZE_FORCE) && cfg->specialize != P25FE_SPECIALIZE_OFF && !(jit_env_off && !must);
    const int plain = rs.dflt ? P25FE_VARIANT_BUILTIN : P25FE_VARIANT_GENERIC;
    if (!want_jit) return plain;
    bool ok = false;
    try {
        t_jit_log.clear();
        ok = p25jit::get_code(jit_spec(rs), jit_dirs(), true, p25jit::default_cache_dir(), false, *code, *path, t_jit_log, from_cache);
    } catch (...) {
        ok = false;
        try { t_jit_log += "exception while specialising\n"; } catch (...) { }
    }
    if (ok) return P25FE_VARIANT_SPECIALIZED;
    if (must) return P25FE_ERR_JIT;
    if (plain == P25FE_VARIANT_GENERIC) fallback_notice();
    return plain;
}

#define HIPCHK(h, expr)                                                        \
    do {                                                                       \
        hipError_t e__ = (expr);                                               \
        if (e__ != hipSuccess) { (h)->last_hip = (int)e__; return P25FE_ERR_HIP; } \
    } while (0)

extern "C" {

void p25fe_default_config(p25fe_config_t* cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->abi_version = P25FE_ABI_VERSION;
    cfg->device = 0;
    cfg->n_channels = 1;
    cfg->n_decim_taps = P25FE_T1;
    cfg->n_chan_taps = P25FE_T2;
    memcpy(cfg->decim_taps, P25FE_DEFAULT_DECIM_TAPS, sizeof(float) * P25FE_T1);
    memcpy(cfg->chan_taps, P25FE_DEFAULT_CHAN_TAPS, sizeof(float) * P25FE_T2);
    cfg->specialize = P25FE_SPECIALIZE_AUTO;
    cfg->fm_deviation_hz = P25FE_FM_DEVIATION_HZ;                    // FmDemod::new(5000, BASEBAND_SAMPLE_RATE), src/demod.rs:54
    cfg->fm_sample_rate_hz = P25FE_FM_SAMPLE_RATE_HZ;
    cfg->fm_gain = 0.0f;                                             // derived from the two
    cfg->u8_scale = P25FE_U8_SCALE;
    cfg->u8_offset = P25FE_U8_OFFSET;
    cfg->u8_lut_valid = 0;
    cfg->decim_phase = P25FE_DECIM_PHASE;                            // Decimator::new(5): the 5th sample of every five, src/demod.rs:50, 87-90
    cfg->n_avg_taps = P25FE_BOXCAR;                                  // MovingAverage::new(10), src/demod.rs:52
    memcpy(cfg->avg_taps, P25FE_DEFAULT_AVG_TAPS, sizeof(float) * P25FE_BOXCAR);
}

const char* p25fe_strerror(int status)
{
    switch (status) {
    case P25FE_OK: return "ok";
    case P25FE_ERR_ARG: return "invalid argument";
    case P25FE_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
    case P25FE_ERR_HIP: return "HIP runtime error";
    case P25FE_ERR_CAPACITY: return "output buffer too small";
    case P25FE_ERR_FORMAT: return "sample format changed within a stream";
    case P25FE_ERR_NOMEM: return "out of memory";
    case P25FE_ERR_JIT: return "kernel specialisation failed (p25fe_specialize_log)";
    case P25FE_ERR_TIMEOUT: return "a device-side wait gave up (time shard: the head segment never arrived)";
    default: return "unknown status";
    }
}

int p25fe_last_hip_error(const p25fe_t* h) { return h ? h->last_hip : 0; }
int p25fe_device(const p25fe_t* h) { return h ? h->cfg.device : -1; }
int p25fe_kernel_variant(const p25fe_t* h) { return h ? h->variant : -1; }

size_t p25fe_specialize_log(char* buf, size_t cap)
{
    if (!buf || cap == 0) return 0;
    const size_t n = t_jit_log.size() < cap - 1 ? t_jit_log.size() : cap - 1;
    memcpy(buf, t_jit_log.data(), n);
    buf[n] = '\0';
    return n;
}

int p25fe_specialize(const p25fe_config_t* cfg, const char* dir, char* path_out, size_t path_cap)
{
    if (!cfg) return P25FE_ERR_ARG;
    if (path_out && path_cap) path_out[0] = '\0';
    Resolved r;
    const int rc = resolve_config(cfg, &r);
    if (rc) return rc;
    if (r.dflt && cfg->specialize != P25FE_SPECIALIZE_FORCE) return P25FE_OK;   // the library's own kernels carry these numbers
    try {
        t_jit_log.clear();
        const std::string d = (dir && *dir) ? std::string(dir) : p25jit::default_cache_dir();
        std::vector<char> code;
        std::string path;
        if (!p25jit::get_code(jit_spec(r), {d}, true, d, /*aot=*/true, code, path, t_jit_log) || path.empty()) return P25FE_ERR_JIT;
        if (path_out && path_cap) snprintf(path_out, path_cap, "%s", path.c_str());
    } catch (...) {
        return P25FE_ERR_NOMEM;
    }
    return P25FE_OK;
}

int p25fe_probe_variant(const p25fe_config_t* cfg)
{
    if (!cfg) return P25FE_ERR_ARG;
    Resolved r;
    const int rc = resolve_config(cfg, &r);
    if (rc) return rc;
    std::vector<char> code;
    std::string path;
    try {
        return choose_variant(cfg, r, &code, &path, nullptr);
    } catch (...) {
        return P25FE_ERR_NOMEM;
    }
}

static inline size_t n_baseband_ph(int phase, uint64_t abs0, size_t n)
{
    const size_t o0 = (size_t)(((uint64_t)phase + 5 - abs0 % 5) % 5);       // first decimation instant inside the range
    return n > o0 ? (n - o0 - 1) / 5 + 1 : 0;
}
size_t p25fe_n_baseband(uint64_t abs0, size_t n) { return n_baseband_ph(P25FE_DECIM_PHASE, abs0, n); }
size_t p25fe_n_baseband_h(const p25fe_t* h, uint64_t abs0, size_t n) { return n_baseband_ph(h ? h->phase : P25FE_DECIM_PHASE, abs0, n); }

size_t p25fe_shard_halo(void) { return SHARD_HALO; }

int p25fe_reset(p25fe_t* h)
{
    if (!h) return P25FE_ERR_ARG;
    h->abs_iq = 0;
    h->fmt_locked = -1;
    h->abs_bb = 0;
    h->rs_idx = nullptr; h->rs_n = 0; h->rs_stride = 0;             // a pending lock-drop list belonged to the old stream
    try {                                                           // no exception crosses the C boundary
        h->anchor.assign((size_t)h->C, p25fe_anchor_t{0, 0.f, 0.f, 0.f, 0, SPS, 1});
        h->total_dibits.assign((size_t)h->C, 0);
        h->hist_iq.assign((size_t)h->C * SHARD_HALO * 8, 0);
        h->tail_bb.assign((size_t)h->C * BBPAD, 0.f);
    } catch (...) {
        return P25FE_ERR_NOMEM;
    }
    return P25FE_OK;
}

int p25fe_create(const p25fe_config_t* cfg, p25fe_t** out)
{
    if (!cfg || !out) return P25FE_ERR_ARG;
    *out = nullptr;
    if (cfg->abi_version != P25FE_ABI_VERSION || cfg->n_channels < 1 || cfg->n_channels > 65535 /* grid.y */) return P25FE_ERR_ARG;
    Resolved rs;
    if (int rrc = resolve_config(cfg, &rs)) return rrc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return P25FE_ERR_NO_DEVICE;
    if (hipSetDevice(cfg->device) != hipSuccess) return P25FE_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return P25FE_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return P25FE_ERR_NO_DEVICE;   // code object is gfx950 only

    p25fe_t* h = new (std::nothrow) p25fe;
    if (!h) return P25FE_ERR_NOMEM;
    h->cfg = *cfg;
    h->C = cfg->n_channels;
    h->track = cfg->symbol_clock & 0xff;
    h->causal_ok = (cfg->symbol_clock & P25FE_CLOCK_CAUSAL_OK) != 0;
    h->look = h->track ? CLK_L : 0;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->taps = rs.taps;
    h->phase = cfg->decim_phase;
    h->n_avg = rs.taps.n_avg;
    h->long_taps = rs.long_taps;
    h->u8_lut_mode = !rs.lut_affine;
    h->variant = rs.dflt ? P25FE_VARIANT_BUILTIN : P25FE_VARIANT_GENERIC;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return P25FE_ERR_HIP; }
    hipError_t e = hipSuccess;
    {
        const char* pv = getenv("P25FE_K1_P");            // tuning knob: FIR outputs per lane (5 default, 3)
        h->k1_p = (pv && atoi(pv) == 3) ? 3 : 5;
        // measurement knobs of the two-stream form: events attached to the dispatches (default) or
        // recorded separately; CUs the receive stream is confined to (0 = no mask)
        const char* ee = getenv("P25FE_EXT_EVENTS");
        h->ext_events = !(ee && atoi(ee) == 0);
        const char* rc_ = getenv("P25FE_RX_CUS");
        h->rx_cus = rc_ ? atoi(rc_) : 0;
        for (int k = 0; k < MAX_LANES - 1; ++k) h->spare[k].id = k + 1;
        const char* pd = getenv("P25FE_SHARD_PIPE_DEPTH");            // (measurement knob: 2 .. MAX_LANES)
        if (pd && atoi(pd) >= 2 && atoi(pd) <= MAX_LANES) h->sh_depth = atoi(pd);
        // The general receiver's chain (tracking clock) is the longer one: 150 - 230 us beside a K1 of 275.  Round 5's held a 512-thread
        // k_scan_g that waited for K1's drain, and a third scratch set was worth 12 % (0.326 -> 0.286 ms per pipelined step); with the chain
        // as one-wave work it still is worth 0.6 - 0.9 % (profiles/r06_run_depth.txt).  The fixed-stride chain: two sets.
        if (cfg->symbol_clock != P25FE_CLOCK_FIXED) h->run_depth = 3;
        const char* rd = getenv("P25FE_PIPE_DEPTH");
        if (rd && atoi(rd) >= 2 && atoi(rd) <= MAX_LANES) h->run_depth = atoi(rd);
    }
    if (e == hipSuccess) e = h->d_taps.ensure(sizeof(Taps));
    if (e == hipSuccess) e = hipMemcpy(h->d_taps.p, &h->taps, sizeof(Taps), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipStreamDestroy(h->stream); h->d_taps.release(); delete h; return P25FE_ERR_HIP; }
    // Numbers other than the build's own: the same kernels with THOSE numbers as immediates (cached code object, else
    // hipRTC -- seconds, once per set of numbers and library build); the generic LDS-tap kernels if that is off or fails.
    {
        std::vector<char> code;
        std::string path;
        bool from_cache = false;
        int var = choose_variant(cfg, rs, &code, &path, &from_cache);
        if (var < 0) { p25fe_destroy(h); return var; }
        if (var == P25FE_VARIANT_SPECIALIZED) {
            bool ok = false;
            try {
                hipError_t me = hipModuleLoadData(&h->jit_mod, code.data());
                if (me != hipSuccess && from_cache) {
                    // a verified file the loader still refuses (another architecture's object under this name): compile afresh;
                    // the stale entry is removed only from the cache -- never from a deployment's $P25FE_SPEC_DIR
                    (void)hipGetLastError();
                    t_jit_log += "cached code object " + path + " did not load: recompiling\n";
                    const std::string store = p25jit::default_cache_dir();
                    if (path.compare(0, store.size() + 1, store + "/") == 0) (void)remove(path.c_str());
                    h->jit_mod = nullptr;
                    if (p25jit::get_code(jit_spec(rs), {}, true, store, false, code, path, t_jit_log)) me = hipModuleLoadData(&h->jit_mod, code.data());
                }
                ok = me == hipSuccess;
                for (int f = 0; f < 2 && ok; ++f)
                    for (int k = 0; k < 3 && ok; ++k)
                        ok = hipModuleGetFunction(&h->jit_fn[f][k], h->jit_mod, p25jit::KERNEL_NAMES[f][k]) == hipSuccess;
                if (!ok) { (void)hipGetLastError(); t_jit_log += "loading the specialised code object failed\n"; }
            } catch (...) {
                ok = false;
            }
            if (!ok) {
                if (h->jit_mod) { (void)hipModuleUnload(h->jit_mod); h->jit_mod = nullptr; }
                if (cfg->specialize >= P25FE_SPECIALIZE_REQUIRE) { p25fe_destroy(h); return P25FE_ERR_JIT; }
                var = rs.dflt ? P25FE_VARIANT_BUILTIN : P25FE_VARIANT_GENERIC;
                if (var == P25FE_VARIANT_GENERIC) fallback_notice();
            }
        }
        h->variant = var;
    }
    int rc = p25fe_reset(h);
    if (rc != P25FE_OK) { p25fe_destroy(h); return rc; }
    *out = h;
    return P25FE_OK;
}

void p25fe_destroy(p25fe_t* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->rx_stream) { (void)hipStreamSynchronize(h->rx_stream); (void)hipStreamDestroy(h->rx_stream); }
    if (h->jit_mod) (void)hipModuleUnload(h->jit_mod);
    if (h->win_cs) { (void)hipStreamSynchronize(h->win_cs); (void)hipStreamDestroy(h->win_cs); }
    if (h->win_os) { (void)hipStreamSynchronize(h->win_os); (void)hipStreamDestroy(h->win_os); }
    for (auto& slot : h->win_ev) for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) { h->win_buf[b].release(); h->win_dib[b].release(); h->win_stage[b].release(); }
    h->win_res.release(); h->win_anc.release(); h->win_out.release();
    for (int l = 0; l < MAX_LANES; ++l) {
        if (h->ev_k1[l]) (void)hipEventDestroy(h->ev_k1[l]);
        if (h->ev_rx[l]) (void)hipEventDestroy(h->ev_rx[l]);
    }
    for (auto& a : h->spare) {
        DevBuf* alt[] = {&a.pl_f, &a.pl_bits, &a.evl, &a.evthr, &a.recs, &a.tsum, &a.outs, &a.gsum, &a.gouts, &a.evg, &a.gagg, &a.gpre, &a.gtick, &a.gsg, &a.gpg};
        for (DevBuf* b : alt) b->release();
    }
    h->gsum.release(); h->gouts.release(); h->evg.release(); h->gsg.release(); h->gpg.release();
    h->evrec.release(); h->evnext.release(); h->evoff.release();
    DevBuf* bufs[] = {&h->pl_f, &h->pl_bits, &h->evl, &h->evthr, &h->recs, &h->tsum, &h->outs, &h->power_partial, &h->chunk_cnt, &h->d_taps, &h->sh_flag,
                      &h->gagg, &h->gpre, &h->gtick};
    for (DevBuf* b : bufs) b->release();
    h->hin.release(); h->hbb.release(); h->hout.release();
    for (auto& e : h->prof_ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

}  // extern "C"

// --------------------------------------------------------------------------------------------
// profiling hook
// --------------------------------------------------------------------------------------------
static void prof_begin(p25fe_t* h)
{
    // level 3: only every PROF_SAMPLE-th call carries events (the two records around K1 are two more packets between
    // consecutive kernels, ~4 us each)
    if (h->prof_on && h->prof_level == 3 && (h->prof_seq++ % PROF_SAMPLE) != 0) { h->prof_slot = -1; return; }
    h->prof_slot = h->prof_on ? (int)(h->prof_calls++ % PROF_RING) : -1;
    if (h->prof_slot >= 0) h->prof_mask[(size_t)h->prof_slot] = 0;
}
static void prof_mark(p25fe_t* h, int idx, hipStream_t st)
{
    if (h->prof_slot >= 0 && (idx <= 1 || h->prof_level == 1)) {    // levels 2, 3: only the two events around K1
        if (idx <= 1 && h->ext_events) return;                      // K1's pair rides on its dispatch: prof_k1_events
        if (hipEventRecord(h->prof_ev[(size_t)h->prof_slot * 5 + idx], st) == hipSuccess)
            h->prof_mask[(size_t)h->prof_slot] |= (uint8_t)(1u << idx);
    }
}
// The pair of events of the current profiling slot for K1's own dispatch (null when this call is not sampled): the
// kernel's begin / end time stamps, no extra packet on the stream.
static void prof_k1_events(p25fe_t* h, hipEvent_t* e0, hipEvent_t* e1)
{
    *e0 = *e1 = nullptr;
    if (h->prof_slot < 0 || !h->ext_events) return;
    *e0 = h->prof_ev[(size_t)h->prof_slot * 5 + 0];
    *e1 = h->prof_ev[(size_t)h->prof_slot * 5 + 1];
    h->prof_mask[(size_t)h->prof_slot] |= 3u;
}

// --------------------------------------------------------------------------------------------
// internal launchers
// --------------------------------------------------------------------------------------------
static int ensure_chunk_scratch(p25fe_t* h)
{
    if (h->chunk_cnt.p) return P25FE_OK;
    HIPCHK(h, h->chunk_cnt.ensure(sizeof(unsigned) * (size_t)h->C));
    HIPCHK(h, hipMemsetAsync(h->chunk_cnt.p, 0, sizeof(unsigned) * (size_t)h->C, h->stream));
    return P25FE_OK;
}
static int launch_frontend(p25fe_t* h, const void* d_x, int fmt, size_t ch_stride, size_t n_hist, size_t n,
                           uint64_t abs0, long m_begin, float* d_bb, size_t bb_stride, float* d_power_dbm,
                           hipStream_t st, const PlanarGeo* planar = nullptr, int part = 0, hipEvent_t ev0 = nullptr,
                           hipEvent_t ev1 = nullptr, const ChunkRecvArgs* chunk = nullptr, long* head_end = nullptr,
                           unsigned* done_flag = nullptr, unsigned done_seq = 0u)
{
    // head_end (nullable, part 2): receives the planar position up to which this launch's segments write
    // done_flag (nullable, planar): the launch's last workgroup publishes done_flag[0] = done_seq (K1Args.done_flag)
    // chunk (nullable, planar only): launch k_chunk -- K1 plus the one-tile receiver run by each channel's last workgroup
    // ev0 / ev1 (nullable): events attached to K1's dispatch (begin / end of the kernel)
    // part: 0 = every segment; 1 = only the segments whose input window lies inside the owned samples (a shard's main
    // launch, runs while the halo is still on the wire); 2 = the others (the shard's head, after the halo has arrived)
    if (fmt != P25FE_FMT_CF32 && fmt != P25FE_FMT_U8) return P25FE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_x) & 15u) != 0) return P25FE_ERR_ARG;     // 16-B vector loads
    if (h->C > 1 && (ch_stride % (fmt == P25FE_FMT_CF32 ? 2 : 8)) != 0) return P25FE_ERR_ARG;
    const bool pro = seg_prologue(fmt);                            // p25fe_kernels.hip: segment prologue (u8) or recomputed halo (cf32)
    // planar output: the first output K1 produces is the form's (a block boundary of the layout, or 80 outputs in front of
    // the receiver's 240-sample history) -- the caller's value is ignored
    if (planar) m_begin = pro ? -(long)PLPAD - h->look : -(long)HIST_BB - h->look;
    const size_t n_out = p25fe_n_baseband_h(h, abs0, n);
    const long total = (long)n_out - m_begin;
    if (total <= 0) {
        // power_dbm of an empty chunk: the reference divides 0 by 0 (src/demod.rs:123-134) -> NaN (0xffffffff is a quiet NaN)
        if (d_power_dbm) HIPCHK(h, hipMemsetAsync(d_power_dbm, 0xff, sizeof(float) * (size_t)h->C, st));
        return P25FE_OK;
    }
    // Segments are SHORT: three sub-tiles per one-wave workgroup.  Measured on config 2 (profiles/): one long
    // segment per resident wave (44 sub-tiles, no tail round) ran K1 in 0.305 ms, 2-4 sub-tiles per workgroup in
    // 0.23-0.27 ms depending on the box -- neighbouring workgroups then stream neighbouring DRAM pages and the
    // dispatcher balances the CUs, which outweighs recomputing the 50-sample filter halo once per segment (5.5 % at
    // 3 sub-tiles; A/B on one box, three rounds: 2 -> 0.275, 3 -> 0.264, 4 -> 0.269 ms).  P25FE_SUBS overrides.
    const int pk = (planar || h->long_taps || h->variant == P25FE_VARIANT_SPECIALIZED) ? 5 : h->k1_p;
    const int t1 = h->long_taps ? TMAX : T1;
    const long sub = (long)WV * pk;
    static const long subs_env = [] { const char* e = getenv("P25FE_SUBS"); return e ? atol(e) : 0L; }();
    // (round 2: the instruction-bound u8 kernel prefers longer segments -- less halo recomputed: 3 -> 239, 6 -> 231 us on one
    // box; with the final build 4 -> 222 / 216, 6 -> 217 / 215, 9 -> 212 / 209, 12 -> 216 / 205 us under rocprofv3)
    long subs = subs_env > 0 ? subs_env : (fmt == P25FE_FMT_U8 ? 9 : 3);
    if (chunk) subs = 1;                                            // a chunk is latency: every sub-tile its own workgroup
    if (subs > 32768) subs = 32768;
    // the post-discriminator filter's length as far as the kernel's GEOMETRY goes (frontend_body's T3): the handle's own for the
    // immediate-coefficient kernels, the ABI's ceiling for the generic ones; the halo form's recomputed halo follows the
    // handle's real length either way (the generic kernels take it as an argument)
    const bool ct_k = h->variant != P25FE_VARIANT_GENERIC;
    const int t2e = h->long_taps ? TMAX : T2;
    const int t3geo = ct_k ? h->n_avg : TMAX;
    const long segh = seg_halo_for(h->n_avg, t2e);
    const long seg_len = pro ? subs * sub : (sub - segh) + (subs - 1) * sub;
    // what a segment needs in front of its first output: the prologue's decimator outputs, or the recomputed halo
    const long nd = pro ? t3geo + t2e - 1 : segh;
    // a time shard's launches (part != 0): the segments in front of the first one whose input lies inside the owned samples
    // are ONE sub-tile long (K1Args.lead_segs) -- the head is then a few one-sub-tile workgroups side by side
    const long lead_len = pro ? sub : sub - segh;
    const long o0l = (long)(((uint64_t)h->phase + 5 - abs0 % 5) % 5);
    long lead = 0;
    if (part && subs > 1 && !chunk)
        while (lead < 64 && o0l + DEC * (m_begin + lead * lead_len - nd) - (t1 - 1) < 0 && lead * lead_len < total) ++lead;
    auto seg_start = [&](long k) { return k < lead ? m_begin + k * lead_len : m_begin + lead * lead_len + (k - lead) * seg_len; };
    const long n_seg = lead * lead_len >= total ? (total + lead_len - 1) / lead_len : lead + (total - lead * lead_len + seg_len - 1) / seg_len;
    const long pl_shift = PLPAD + h->look;       // the general receiver sees the range h->look samples late (p25fe_recv.hip)
    if (planar && pro && (m_begin + pl_shift < 0 || (m_begin + pl_shift) % PL_BLK != 0 || seg_len % PL_BLK != 0)) return P25FE_ERR_ARG;
    // (halo form: the outputs a segment recomputes and drops may lie in front of planar position 0 -- with a 160-output halo the
    // range's first segment starts 80 positions in front of it; they are never stored, only whole bytes of the planes are)
    if (planar && !pro && (m_begin + pl_shift < 0 || (m_begin + pl_shift) % 80 != 0 || seg_len % 80 != 0)) return P25FE_ERR_ARG;
    if (planar && n_out > MAX_RANGE_BB) return P25FE_ERR_ARG;

    K1Args a;
    a.x = d_x;
    a.ch_stride = (long)ch_stride;
    a.n_hist = (long)n_hist;
    a.n_new = (long)n;
    a.o0 = (int)o0l;
    a.bb = d_bb;
    a.bb_stride = (long)bb_stride;
    a.n_out = (long)n_out;
    a.subs_per_seg = (int)subs;
    a.seg_first = 0;
    long seg_count = n_seg;
    if (part) {
        if (d_power_dbm) return P25FE_ERR_ARG;
        // segment k reads input from o0 + 5 (its first output - nd) - (T1 - 1) on
        long k_min = 0;
        while (k_min < n_seg && o0l + DEC * (seg_start(k_min) - nd) - (t1 - 1) < 0) ++k_min;
        if (part == 1) { a.seg_first = (int)k_min; seg_count = n_seg - k_min; }
        else seg_count = k_min;
        if (head_end) *head_end = seg_start(k_min) - m_begin;
        if (seg_count <= 0) return P25FE_OK;
    }
    a.seg_count = (int)seg_count;
    a.n_ch = h->C;
    a.m_begin = m_begin;
    a.power_partial = nullptr;
    a.bbp = nullptr; a.bbp_ch_stride = 0; a.bits = nullptr; a.bits_ch_stride = 0; a.pl_shift = (int)pl_shift;
    a.done_flag = planar ? done_flag : nullptr; a.done_seq = done_seq;
    a.lead_segs = (int)lead;
    a.seg_halo = (int)segh;
    if (planar) {
        a.bbp = h->pl_f.as<float>(); a.bbp_ch_stride = (long)planar->floats();
        a.bits = h->pl_bits.as<uint8_t>(); a.bits_ch_stride = (long)(4 * planar->words());
    }
    if (d_power_dbm) {
        HIPCHK(h, h->power_partial.ensure(sizeof(float) * (size_t)h->C * (size_t)n_seg));
        a.power_partial = h->power_partial.as<float>();
    }
    const long n_items = seg_count * (long)h->C;
    if (n_items > 0x7fffffffL) return P25FE_ERR_ARG;
    const Taps* dt = h->d_taps.as<Taps>();
    const bool u8 = fmt == P25FE_FMT_U8;
    const dim3 grid((unsigned)seg_count, (unsigned)h->C);            // one one-wave workgroup per (segment, channel)
    // LDS per workgroup: [d carry | window] (+ taps: generic kernels) (+ the u8 table: generic u8 kernels, and specialised ones
    // whose table is not affine).  Immediate-coefficient kernels never touch the taps area at the end of the layout: not
    // allocated.  (13 376 B per wave is 11 waves per CU; a 12th would need 13 312 -- trimming to that changed nothing.)
    const bool ct = h->variant != P25FE_VARIANT_GENERIC;
    const bool lut = u8 && (!ct || h->u8_lut_mode);
    size_t lds = h->long_taps ? k1_lds_bytes<Geo<5, 1>>(ct, lut, t3geo) : (pk == 3 ? k1_lds_bytes<Geo<3>>(ct, lut, t3geo) : k1_lds_bytes<Geo<5>>(ct, lut, t3geo));
    // (experiments: extra dynamic LDS per workgroup = fewer resident waves per CU; the occupancy sensitivity of docs/MEASUREMENTS.md)
    static const size_t lds_pad_env = [] { const char* e = getenv("P25FE_K1_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
    lds += lds_pad_env;
    ChunkTail tail;
    if (chunk) {
        if (!planar || part) return P25FE_ERR_ARG;
        int rc = ensure_chunk_scratch(h);
        if (rc) return rc;
        tail.r = *chunk; tail.counter = h->chunk_cnt.as<unsigned>(); tail.wg_per_ch = (int)seg_count;
    }
    if (h->variant == P25FE_VARIANT_SPECIALIZED) {
        // kernels compiled for this handle's numbers (p25fe_jit.cpp): same source, same launch shape, C entry points
        hipFunction_t f = h->jit_fn[u8 ? 1 : 0][chunk ? 2 : (planar ? 1 : 0)];
        void* params[3] = {&a, &dt, &tail};
        if (ev0 || ev1)
            HIPCHK(h, hipExtModuleLaunchKernel(f, grid.x * WV, grid.y, 1, WV, 1, 1, lds, st, params, nullptr, ev0, ev1, 0));
        else
            HIPCHK(h, hipModuleLaunchKernel(f, grid.x, grid.y, 1, WV, 1, 1, (unsigned)lds, st, params, nullptr));
    } else {
#define P25FE_K1_CASE(FMT, CT, PK, OM, TX) launch_ev(k_frontend<FMT, CT, PK, OM, TX>, grid, dim3(WV), lds, st, ev0, ev1, a, dt)
#define P25FE_CHUNK_CASE(FMT, CT, TX) hipLaunchKernelGGL((k_chunk<FMT, CT, TX>), grid, dim3(WV), lds, st, a, dt, tail)
        if (chunk) {
            if (h->long_taps) { if (u8) P25FE_CHUNK_CASE(P25FE_FMT_U8, false, 1); else P25FE_CHUNK_CASE(P25FE_FMT_CF32, false, 1); }
            else if (ct) { if (u8) P25FE_CHUNK_CASE(P25FE_FMT_U8, true, 0); else P25FE_CHUNK_CASE(P25FE_FMT_CF32, true, 0); }
            else { if (u8) P25FE_CHUNK_CASE(P25FE_FMT_U8, false, 0); else P25FE_CHUNK_CASE(P25FE_FMT_CF32, false, 0); }
        } else if (h->long_taps) {
            if (planar) { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, false, 5, OUT_PLANAR, 1); else P25FE_K1_CASE(P25FE_FMT_CF32, false, 5, OUT_PLANAR, 1); }
            else { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, false, 5, OUT_LINEAR, 1); else P25FE_K1_CASE(P25FE_FMT_CF32, false, 5, OUT_LINEAR, 1); }
        } else if (planar) {
            if (ct) { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, true, 5, OUT_PLANAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, true, 5, OUT_PLANAR, 0); }
            else { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, false, 5, OUT_PLANAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, false, 5, OUT_PLANAR, 0); }
        } else if (pk == 3) {
            if (ct) { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, true, 3, OUT_LINEAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, true, 3, OUT_LINEAR, 0); }
            else { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, false, 3, OUT_LINEAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, false, 3, OUT_LINEAR, 0); }
        } else {
            if (ct) { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, true, 5, OUT_LINEAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, true, 5, OUT_LINEAR, 0); }
            else { if (u8) P25FE_K1_CASE(P25FE_FMT_U8, false, 5, OUT_LINEAR, 0); else P25FE_K1_CASE(P25FE_FMT_CF32, false, 5, OUT_LINEAR, 0); }
        }
#undef P25FE_K1_CASE
#undef P25FE_CHUNK_CASE
    }
    if (chunk) { HIPCHK(h, hipGetLastError()); return P25FE_OK; }
    HIPCHK(h, hipGetLastError());
    if (d_power_dbm) {
        hipLaunchKernelGGL(k_power_finish, dim3((unsigned)h->C), dim3(256), 0, st, a.power_partial, (int)n_seg,
                           (long)n_out, d_power_dbm);
        HIPCHK(h, hipGetLastError());
    }
    return P25FE_OK;
}

static int ensure_slice_scratch(p25fe_t* h, size_t n_bb)
{
    const size_t C = (size_t)h->C;
    const PlanarGeo g(n_bb);
    HIPCHK(h, h->pl_f.ensure(C * g.floats() * sizeof(float)));
    HIPCHK(h, h->pl_bits.ensure(C * g.words() * sizeof(uint32_t)));
    HIPCHK(h, h->evl.ensure(C * g.n_tiles * EVCAP * sizeof(uint16_t)));
    HIPCHK(h, h->evthr.ensure(C * g.n_tiles * EVTHR_N * 3 * sizeof(float)));
    HIPCHK(h, h->recs.ensure(C * g.n_tiles * sizeof(TileRec)));
    HIPCHK(h, h->tsum.ensure(C * g.n_tiles * sizeof(unsigned long long)));
    HIPCHK(h, h->outs.ensure(C * g.n_tiles * sizeof(ScanOut)));
    const size_t ng = (size_t)n_groups_of((int)g.n_tiles);
    {
        HIPCHK(h, h->gagg.ensure(C * ng * sizeof(GroupAgg)));
        HIPCHK(h, h->gpre.ensure(C * ng * sizeof(GroupPre)));
        const void* before = h->gtick.p;
        HIPCHK(h, h->gtick.ensure(C * (ng + 1) * sizeof(unsigned)));
        if (h->gtick.p != before) HIPCHK(h, hipMemset(h->gtick.p, 0, h->gtick.cap));    // a fresh buffer: the counters start at zero and return to it
    }
    if (h->track || h->rs_n) {                                       // the general receiver's summaries and carry-ins
        HIPCHK(h, h->gsum.ensure(C * g.n_tiles * sizeof(TileSumG)));
        HIPCHK(h, h->gouts.ensure(C * g.n_tiles * sizeof(ScanOutG)));
        HIPCHK(h, h->evg.ensure(C * g.n_tiles * EVCAP * sizeof(uint32_t)));
        HIPCHK(h, h->gsg.ensure(C * ng * sizeof(GroupSumG)));
        HIPCHK(h, h->gpg.ensure(C * ng * sizeof(GroupPreG)));
    }
    return P25FE_OK;
}

// The receiver options of the call being enqueued.  h->rs_* (p25fe_resync_at_dev) is consumed by the call that runs the
// sync detection; a shard keeps its copy for pass 2.
struct RecvCall {
    bool gen = false;
    bool reslice = false;       // SPEC 3.8c: set by the entry points that hold the whole range (p25fe_run_dev*, p25fe_slice_dev)
    RecvOpt opt;
};
static RecvCall recv_call(const p25fe_t* h)
{
    RecvCall c;
    c.opt.track = h->track;
    c.opt.n_resync = (int)h->rs_n;
    c.opt.resync = h->rs_idx;
    c.opt.resync_stride = (long)h->rs_stride;
    c.gen = h->track != 0 || h->rs_n != 0;
    return c;
}

static Planar planar_view(const p25fe_t* h, const PlanarGeo& g)
{
    Planar p;
    p.f = h->pl_f.as<float>(); p.f_ch = (long)g.floats();
    p.bits = h->pl_bits.as<uint32_t>(); p.bits_ch = (long)g.words();
    return p;
}

// linear baseband (n_hist_bb valid samples before d_bb) -> the planar scratch
static int launch_planarize(p25fe_t* h, const float* d_bb, size_t bb_stride, size_t n_hist_bb, size_t n_bb, hipStream_t st)
{
    const PlanarGeo g(n_bb);
    PlanarizeArgs a;
    a.bb = d_bb; a.bb_stride = (long)bb_stride; a.n_hist = (long)n_hist_bb; a.n = (long)n_bb;
    a.f = h->pl_f.as<float>(); a.f_ch = (long)g.floats();
    a.bits = h->pl_bits.as<uint32_t>(); a.bits_ch = (long)g.words(); a.n_blocks = (long)g.n_blocks;
    a.shift = (int)h->look;
    hipLaunchKernelGGL(k_planarize, dim3((unsigned)((g.n_blocks + 1) / 2), (unsigned)h->C), dim3(WV * SPS), 0, st, a);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

// K3's arguments, fixed-stride receiver (k_scan_tiles / k_range_scan)
static void scan_args(p25fe_t* h, size_t n_bb, long abs_bb0, int n_tiles, const p25fe_anchor_t* d_anchor_in, p25fe_result_t* d_result, ScanArgs* t)
{
    t->tsum = h->tsum.as<unsigned long long>(); t->recs = h->recs.as<TileRec>(); t->n_tiles = n_tiles; t->n = (long)n_bb; t->abs0 = abs_bb0;
    t->outs = h->outs.as<ScanOut>(); t->gagg = h->gagg.as<GroupAgg>(); t->gpre = h->gpre.as<GroupPre>();
    t->tickets = h->gtick.as<unsigned>(); t->anchor_in = d_anchor_in; t->result = d_result;
    t->n_baseband = n_bb;
}
// K2 on the planar scratch (abs_bb0: absolute index of the first PROCESSED sample = owned sample 0 minus h->look)
static int launch_detect(p25fe_t* h, size_t n_bb, long abs_bb0, hipStream_t st, const RecvCall& rc, bool wait_head_flag = false)
{
    const PlanarGeo g(n_bb);
    DetArgs d;
    d.head_flag = nullptr; d.head_seq = 0u; d.head_tile_max = -1; d.head_err = nullptr;
    if (wait_head_flag) {
        d.head_flag = h->sh_flag.as<unsigned>(); d.head_seq = h->sh_seq; d.head_tile_max = h->sh_head_tile_max;
        d.head_err = h->sh_flag.as<unsigned>() + SH_FLAG_ERR;
    }
    d.pl = planar_view(h, g); d.n = (long)n_bb; d.abs0 = abs_bb0; d.n_tiles = (int)g.n_tiles;
    d.recs = h->recs.as<TileRec>(); d.tsum = h->tsum.as<unsigned long long>(); d.evl = h->evl.as<uint16_t>(); d.evthr = h->evthr.as<float>();
    d.opt = rc.opt; d.gsum = h->gsum.as<TileSumG>(); d.evg = h->evg.as<uint32_t>();
    const dim3 grid((unsigned)g.n_tiles, (unsigned)h->C);
    if (rc.gen) hipLaunchKernelGGL(k_detect<true>, grid, dim3(WV), 0, st, d);
    else hipLaunchKernelGGL(k_detect<false>, grid, dim3(WV), 0, st, d);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

// What follows K2: the slicer (do_slice), in front of it whatever completes the per-tile carry-ins.
//   scanned   K2 has just run in this call, for THIS carry-in: K3 is launched whole (k_scan_tiles / k_scan_tiles_g: group scans, then the
//             range's).  Otherwise the group summaries in the scratch are re-scanned under d_anchor_in by the top step alone
//             (k_range_scan / k_range_scan_g: a time shard's pass 2 under resolved anchors; an empty range, whose record is the
//             carry-in handed through).
//   ev_done   (nullable) attached to the LAST kernel this function launches (its completion = the receive side is done)
//   d_dibits2 (nullable) second destination of the dibits
//   fix       (nullable, fixed-stride receiver only) pass 2 of a time shard on pass 1's scan: the slicer applies the shard's carry-in in
//             closed form (ShardFix in p25fe_recv.hip) on top of the group's
static int launch_scan_slice(p25fe_t* h, size_t n_bb, long abs_bb0, const p25fe_anchor_t* d_anchor_in,
                             uint8_t* d_dibits, size_t dibit_stride, int64_t* d_sync_pos, uint64_t* d_sync_dibit,
                             size_t sync_stride, p25fe_result_t* d_result, bool do_slice, hipStream_t st,
                             const RecvCall& rc, bool scanned, hipEvent_t ev_done = nullptr, uint8_t* d_dibits2 = nullptr,
                             const ShardFix* fix = nullptr)
{
    const PlanarGeo g(n_bb);
    const int n_tiles = n_bb ? (int)g.n_tiles : 0;
    const bool slice = do_slice && n_tiles != 0;
    bool ev_pending = ev_done != nullptr;                            // no kernel has carried ev_done yet
    if (rc.gen) {
        ScanArgsG c;
        c.gsum = h->gsum.as<TileSumG>(); c.recs = h->recs.as<TileRec>(); c.outs = h->gouts.as<ScanOutG>();
        c.gsg = h->gsg.as<GroupSumG>(); c.gpg = h->gpg.as<GroupPreG>(); c.tickets = h->gtick.as<unsigned>();
        c.n_tiles = n_tiles; c.n = (long)n_bb; c.abs0 = abs_bb0; c.anchor_in = d_anchor_in; c.result = d_result;
        c.n_baseband = n_bb; c.track = h->track;
        if (!scanned) {
            launch_ev(k_range_scan_g, dim3((unsigned)h->C), dim3(WV), 0, st, nullptr, slice ? nullptr : ev_done, c);
            HIPCHK(h, hipGetLastError());
            if (!slice) ev_pending = false;
        } else if (n_tiles) {
            launch_ev(k_scan_tiles_g, dim3((unsigned)n_groups_of(n_tiles), (unsigned)h->C), dim3(WV), 0, st, nullptr, slice ? nullptr : ev_done, c);
            HIPCHK(h, hipGetLastError());
            if (!slice) ev_pending = false;
        }
        prof_mark(h, 3, st);
        if (!slice) {
            if (ev_pending) HIPCHK(h, hipEventRecord(ev_done, st));  // (K2 was the last kernel: the event goes behind it)
            prof_mark(h, 4, st);
            return P25FE_OK;
        }
        // pass B: the tiles' carry-ins under their groups' (one wave per group)
        hipLaunchKernelGGL(k_scan_g_groups, dim3((unsigned)n_groups_of(n_tiles), (unsigned)h->C), dim3(WV), 0, st, c);
        HIPCHK(h, hipGetLastError());
        if (rc.reslice) {
            // SPEC 3.8c: the slicer by detection (k_ev_collect / k_ev_count / k_ev_scan / k_ev_slice) on those carry-ins
            const size_t C = (size_t)h->C;
            size_t cap_ev = (size_t)n_tiles * EVCAP;
            const size_t bound = n_bb / (size_t)(W + 1) + (size_t)n_tiles + 8;
            if (bound < cap_ev) cap_ev = bound;
            const size_t stride = cap_ev + 2;
            HIPCHK(h, h->evrec.ensure(C * stride * sizeof(EvRec)));
            HIPCHK(h, h->evoff.ensure(C * (stride + 1) * sizeof(unsigned long long)));
            {
                const void* before = h->evnext.p;
                HIPCHK(h, h->evnext.ensure(C * stride * sizeof(EvNext)));
                if (h->evnext.p != before) HIPCHK(h, hipMemsetAsync(h->evnext.p, 0, h->evnext.cap, st));   // sequence numbers start above 0
            }
            EvArgs e;
            e.pl = planar_view(h, g); e.n = (long)n_bb; e.abs0 = abs_bb0; e.n_tiles = n_tiles;
            e.outs = h->gouts.as<ScanOutG>(); e.gsum = h->gsum.as<TileSumG>(); e.evl = h->evl.as<uint16_t>(); e.evg = h->evg.as<uint32_t>();
            e.evthr = h->evthr.as<float>(); e.anchor_in = d_anchor_in;
            e.rec = h->evrec.as<EvRec>(); e.nxt = h->evnext.as<EvNext>(); e.off = h->evoff.as<unsigned long long>();
            e.ev_stride = (long)stride; e.seq = ++h->ev_seq; e.result = d_result;
            e.dibits = d_dibits; e.dibit_stride = (long)dibit_stride;
            e.sync_pos = (d_sync_pos && d_sync_dibit) ? d_sync_pos : nullptr; e.sync_dibit = d_sync_dibit; e.sync_stride = (long)sync_stride;
            hipLaunchKernelGGL(k_ev_collect, dim3((unsigned)n_tiles, (unsigned)h->C), dim3(WV), 0, st, e);
            HIPCHK(h, hipGetLastError());
            hipLaunchKernelGGL(k_ev_count, dim3((unsigned)(n_tiles < 1024 ? n_tiles : 1024), (unsigned)h->C), dim3(WV), 0, st, e);
            HIPCHK(h, hipGetLastError());
            hipLaunchKernelGGL(k_ev_scan, dim3((unsigned)h->C), dim3(WV), 0, st, e);
            HIPCHK(h, hipGetLastError());
            // (a tracked period is within 1 / 1024 of the nominal one over long intervals and at least 9 samples over the shortest)
            const size_t max_dibits = n_bb / (SPS - 1) + (size_t)n_tiles + 64;
            const size_t lim = dibit_stride < max_dibits ? dibit_stride : max_dibits;
            launch_ev(k_ev_slice, dim3((unsigned)((lim + WV * 4 - 1) / (WV * 4)), (unsigned)h->C), dim3(WV), 0, st, nullptr, ev_done, e);
            HIPCHK(h, hipGetLastError());
            prof_mark(h, 4, st);
            return P25FE_OK;
        }
        SliceArgsG l;
        l.pl = planar_view(h, g); l.n = (long)n_bb; l.abs0 = abs_bb0; l.n_tiles = n_tiles;
        l.outs = h->gouts.as<ScanOutG>(); l.gsum = h->gsum.as<TileSumG>(); l.recs = h->recs.as<TileRec>();
        l.evl = h->evl.as<uint16_t>(); l.evg = h->evg.as<uint32_t>(); l.evthr = h->evthr.as<float>(); l.anchor_in = d_anchor_in;
        l.dibits = d_dibits; l.dibit_stride = (long)dibit_stride;
        l.sync_pos = (d_sync_pos && d_sync_dibit) ? d_sync_pos : nullptr; l.sync_dibit = d_sync_dibit;
        l.sync_stride = (long)sync_stride; l.track = h->track; l.dibits2 = d_dibits2;
        launch_ev(k_slice_g, dim3((unsigned)n_tiles, (unsigned)h->C), dim3(WV), 0, st, nullptr, ev_done, l);
        HIPCHK(h, hipGetLastError());
        prof_mark(h, 4, st);
        return P25FE_OK;
    }
    if (!fix) {
        ScanArgs c;
        scan_args(h, n_bb, abs_bb0, n_tiles, d_anchor_in, d_result, &c);
        // K2 has just run: the whole scan (groups, then the range); otherwise only the range's, on the group aggregates that are there
        if (scanned && n_tiles) launch_ev(k_scan_tiles, dim3((unsigned)n_groups_of(n_tiles), (unsigned)h->C), dim3(WV), 0, st, nullptr, slice ? nullptr : ev_done, c);
        else launch_ev(k_range_scan, dim3((unsigned)h->C), dim3(WV), 0, st, nullptr, slice ? nullptr : ev_done, c);
        HIPCHK(h, hipGetLastError());
        if (!slice) ev_pending = false;
    }
    prof_mark(h, 3, st);
    if (!slice) {
        if (ev_pending) HIPCHK(h, hipEventRecord(ev_done, st));      // (K2 was the last kernel: the event goes behind it)
        prof_mark(h, 4, st);
        return P25FE_OK;
    }
    SliceArgs l;
    l.pl = planar_view(h, g); l.n = (long)n_bb; l.abs0 = abs_bb0; l.n_tiles = n_tiles;
    l.outs = h->outs.as<ScanOut>(); l.recs = h->recs.as<TileRec>(); l.tsum = h->tsum.as<unsigned long long>();
    l.evl = h->evl.as<uint16_t>(); l.evthr = h->evthr.as<float>(); l.anchor_in = d_anchor_in;
    l.dibits = d_dibits; l.dibit_stride = (long)dibit_stride;
    l.sync_pos = (d_sync_pos && d_sync_dibit) ? d_sync_pos : nullptr; l.sync_dibit = d_sync_dibit;
    l.sync_stride = (long)sync_stride;
    l.dibits2 = d_dibits2;
    l.gpre = h->gpre.as<GroupPre>();
    if (fix) l.fix = *fix; else memset(&l.fix, 0, sizeof l.fix);
    // (pass 2 of a time shard: one extra workgroup runs the combine for the record beside the slicing ones)
    launch_ev(k_slice, dim3((unsigned)n_tiles + (fix ? 1u : 0u), (unsigned)h->C), dim3(WV), 0, st, nullptr, ev_done, l);
    HIPCHK(h, hipGetLastError());
    prof_mark(h, 4, st);
    return P25FE_OK;
}

// SPEC 3.8c needs the whole range in one call.  The calls that see the stream in pieces (host streaming chunks, host windows, the
// passes of a time shard) can only run 3.8b's causal rule; a handle that asked for 3.8c gets that ONLY if it said so
// (P25FE_CLOCK_CAUSAL_OK) -- otherwise the call is refused instead of quietly returning another receiver's dibits.
static inline bool piecewise_refused(const p25fe_t* h) { return h->track == P25FE_CLOCK_TRACKING_RESLICE && !h->causal_ok; }

// every entry point that overwrites the receiver's scratch: a shard's pass-1 context is gone
static void shard_invalidate(p25fe_t* h)
{
    h->sh_valid = false;
    h->sh_main_nbb = (size_t)-1;             // a later p25fe_shard_pass1_finish must not pair with a main launch whose planes are gone
    h->sh_head_done = false; h->sh_head_flagged = false;
}

static int pipe_join(p25fe_t* h, hipStream_t st);
// stages 6-7 on a LINEAR device baseband: planarize, detect, scan, slice
static int dev_slice(p25fe_t* h, const float* d_bb, size_t bb_stride, size_t n_hist_bb, size_t n_bb,
                     uint64_t abs_bb0, const p25fe_anchor_t* d_anchor_in, uint8_t* d_dibits,
                     size_t dibit_stride, int64_t* d_sync_pos, uint64_t* d_sync_dibit, size_t sync_stride,
                     p25fe_result_t* d_result, hipStream_t st)
{
    if (n_bb > MAX_RANGE_BB) return P25FE_ERR_ARG;
    shard_invalidate(h);
    if (int jrc = pipe_join(h, st)) return jrc;
    const long view0 = (long)abs_bb0 - h->look;          // first processed index: the tracking clock runs h->look samples late
    int rc = ensure_slice_scratch(h, n_bb ? n_bb : 1);
    if (rc) return rc;
    RecvCall rcall = recv_call(h);
    rcall.reslice = h->track == P25FE_CLOCK_TRACKING_RESLICE;      // the whole range is in memory: SPEC 3.8c applies
    h->rs_n = 0;                                         // the lock drops belong to this call
    if (n_bb == 0)        // empty range: only the scan runs (zero tiles) and hands the anchor through
        return launch_scan_slice(h, 0, view0, d_anchor_in, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result,
                                 false, st, rcall, false);
    rc = launch_planarize(h, d_bb, bb_stride, n_hist_bb, n_bb, st);
    if (rc) return rc;
    rc = launch_detect(h, n_bb, view0, st, rcall, false);
    if (rc) return rc;
    prof_mark(h, 2, st);
    return launch_scan_slice(h, n_bb, view0, d_anchor_in, d_dibits, dibit_stride, d_sync_pos, d_sync_dibit,
                             sync_stride, d_result, true, st, rcall, true);
}

extern "C" {

// --------------------------------------------------------------------------------------------
// device-resident ranges
// --------------------------------------------------------------------------------------------
int p25fe_demod_dev(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                    float* d_bb, size_t bb_stride, float* d_power_dbm, void* stream)
{
    if (!h || !d_iq || !d_bb) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return launch_frontend(h, d_iq, fmt, ch_stride, n_hist, n, abs0, 0, d_bb, bb_stride, d_power_dbm,
                           (hipStream_t)stream);
}

size_t p25fe_n_predecim(uint64_t abs0, size_t n)
{
    const size_t o0 = (size_t)((PD - 1 + PD - abs0 % PD) % PD);
    return n > o0 ? (n - o0 - 1) / PD + 1 : 0;
}

int p25fe_predecim_dev(p25fe_t* h, const float* d_iq, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                       float* d_out, size_t out_stride, void* stream)
{
    if (!h || !d_iq || !d_out) return P25FE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15u) != 0 || (h->C > 1 && (ch_stride & 1))) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t n_out = p25fe_n_predecim(abs0, n);
    if (n_out == 0) return P25FE_OK;
    K0Args a;
    a.x = d_iq; a.ch_stride = (long)ch_stride; a.n_hist = (long)n_hist; a.n_new = (long)n;
    a.o0 = (int)((PD - 1 + PD - abs0 % PD) % PD);
    a.y = d_out; a.y_stride = (long)out_stride; a.n_out = (long)n_out;
    const size_t per_wg = (size_t)K0_SUB * K0_SUBS;
    dim3 grid((unsigned)((n_out + per_wg - 1) / per_wg), (unsigned)h->C);
    hipLaunchKernelGGL(k_predecim, grid, dim3(WV), 0, (hipStream_t)stream, a);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_channelise_dev(p25fe_t* h, const float* d_iq, size_t n_hist, size_t n, uint64_t abs0, float* d_out,
                         size_t out_stride, void* stream)
{
    static_assert(P25FE_CHZ_CHANNELS_ABI == CZ_M, "header and spec disagree");
    if (!h || !d_iq || !d_out) return P25FE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15u) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 7u) != 0) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t n_out = p25fe_n_predecim(abs0, n);
    if (n_out == 0) return P25FE_OK;
    if (out_stride < round_up(n_out, (size_t)WV)) return P25FE_ERR_ARG;      // rows are written in whole 64-instant tiles
    ChzArgs a;
    a.x = d_iq; a.n_hist = (long)n_hist; a.n_new = (long)n; a.abs0 = (long)abs0;
    a.o0 = (int)((PD - 1 + PD - abs0 % PD) % PD);
    a.y = d_out; a.y_stride = (long)out_stride; a.n_out = (long)n_out;
    hipLaunchKernelGGL(k_channelise, dim3((unsigned)((n_out + WV - 1) / WV)), dim3(WV), 0, (hipStream_t)stream, a);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_slice_dev(p25fe_t* h, const float* d_bb, size_t bb_stride, size_t n_hist_bb, size_t n_bb, uint64_t abs_bb0,
                    const p25fe_anchor_t* d_anchor_in, uint8_t* d_dibits, size_t dibit_stride, int64_t* d_sync_pos,
                    uint64_t* d_sync_dibit, size_t sync_stride, p25fe_result_t* d_result, void* stream)
{
    if (!h || !d_bb || !d_dibits || !d_result) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return dev_slice(h, d_bb, bb_stride, n_hist_bb, n_bb, abs_bb0, d_anchor_in, d_dibits, dibit_stride, d_sync_pos,
                     d_sync_dibit, sync_stride, d_result, (hipStream_t)stream);
}

// Every entry point that uses the receiver's scratch first makes its stream wait for receive kernels that
// p25fe_run_dev_pipelined left running on the handle's own stream (no-op when nothing is pending).
static int pipe_join(p25fe_t* h, hipStream_t st)
{
    // The events stay pending until the lane is reused: a later call on ANOTHER stream must wait too (the receive
    // kernels may still be running and share the scratch with whatever that call launches).
    for (int l = 0; l < MAX_LANES; ++l)
        if (h->rx_pending[l] && !(h->rx_joined_any[l] && h->rx_joined[l] == st)) {
            HIPCHK(h, hipStreamWaitEvent(st, h->ev_rx[l], 0));
            h->rx_joined[l] = st; h->rx_joined_any[l] = true;
        }
    return P25FE_OK;
}

int p25fe_join_dev(p25fe_t* h, void* stream)
{
    if (!h) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return pipe_join(h, (hipStream_t)stream);
}

// The preamble of a pipelined call: the receive stream and its events exist, the OTHER scratch set becomes the current one, and
// `st` (where K1 is about to overwrite that set's planes) waits for the receive kernels that last read it, two calls back.
static int ensure_rx_stream(p25fe_t* h)
{
    if (!h->rx_stream) {
        if (h->rx_cus > 0) {
            // The receive kernels are a few thousand short one-wave workgroups and one single-workgroup scan.  Left free they
            // take wave slots, LDS and memory-pipe time from K1 on EVERY CU (K1 measured 12 % slower under overlap in round
            // 2's driver run); confined to a few CUs they cost K1 at most that share of the chip while they run.
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int ncu = h->n_cu < 256 ? h->n_cu : 256;
            const int want = h->rx_cus < ncu ? h->rx_cus : ncu;
            // spread over the XCDs: CU c of the mask numbering sits on XCD c % 8 (workgroups are handed out the same way)
            for (int k = 0; k < want; ++k) mask[k / 32] |= 1u << (k % 32);
            if (hipExtStreamCreateWithCUMask(&h->rx_stream, (uint32_t)((ncu + 31) / 32), mask) != hipSuccess) {
                (void)hipGetLastError();
                h->rx_stream = nullptr;
            }
        }
        if (!h->rx_stream) {
            int prio_lo = 0, prio_hi = 0;
            // (highest priority; normal and lowest were measured in round 6 and change nothing: 0.2647 - 0.2656 / 0.2647 - 0.2650 / 0.2651 -
            // 0.2709 ms per step, docs/MEASUREMENTS.md)
            if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
                hipStreamCreateWithPriority(&h->rx_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
                (void)hipGetLastError();
                h->rx_stream = nullptr;
                HIPCHK(h, hipStreamCreateWithFlags(&h->rx_stream, hipStreamNonBlocking));     // any non-blocking stream will do
            }
        }
    }
    return P25FE_OK;
}
static int pipe_open(p25fe_t* h, hipStream_t st, int depth = 2)
{
    if (int rc = ensure_rx_stream(h)) return rc;
    for (int l = 0; l < MAX_LANES; ++l) {                            // (per event: a failed creation is retried by the next call)
        if (!h->ev_k1[l]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_k1[l], hipEventDisableTiming));
        if (!h->ev_rx[l]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rx[l], hipEventDisableTiming));
    }
    // the other scratch set becomes the current one; it was last read by the receive kernels of the call before the
    // previous one, which this call's K1 (it overwrites the planes) has to wait for
    {
        p25fe::RxSet& a = h->spare[0];
        std::swap(h->pl_f, a.pl_f); std::swap(h->pl_bits, a.pl_bits); std::swap(h->evl, a.evl);
        std::swap(h->evthr, a.evthr); std::swap(h->recs, a.recs); std::swap(h->tsum, a.tsum);
        std::swap(h->outs, a.outs);
        std::swap(h->gagg, a.gagg); std::swap(h->gpre, a.gpre); std::swap(h->gtick, a.gtick);
        std::swap(h->gsum, a.gsum); std::swap(h->gouts, a.gouts); std::swap(h->evg, a.evg);
        std::swap(h->gsg, a.gsg); std::swap(h->gpg, a.gpg);
        std::swap(h->lane, a.id);
        // the set just retired goes behind the other spare ones of this depth (depth 2: a plain swap)
        for (int k = 0; k + 2 < depth && k + 1 < MAX_LANES - 1; ++k) std::swap(h->spare[k], h->spare[k + 1]);
    }
    const int lane = h->lane;
    if (h->rx_pending[lane]) {
        const bool nowait = P25FE_M_PIPE_NOWAIT();                   // (false in the product: measurement hook, p25fe_kernels.hip)
        if (!nowait && !(h->rx_joined_any[lane] && h->rx_joined[lane] == st)) HIPCHK(h, hipStreamWaitEvent(st, h->ev_rx[lane], 0));
        h->rx_pending[lane] = false;
    }
    h->rx_joined_any[lane] = false;
    return P25FE_OK;
}

int p25fe_run_dev_pipelined(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n, uint8_t* d_dibits,
                            size_t dibit_stride, p25fe_result_t* d_result, void* stream)
{
    if (!h || !d_iq || !d_dibits || !d_result || p25fe_n_baseband_h(h, 0, n) > MAX_RANGE_BB) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    shard_invalidate(h);
    if (int orc = pipe_open(h, st, h->run_depth)) return orc;
    const int lane = h->lane;
    const size_t n_bb = p25fe_n_baseband_h(h, 0, n);
    int rc = P25FE_OK;
    const PlanarGeo g(n_bb);
    hipEvent_t k1_done = h->ev_k1[lane];
    bool k1_done_attached = false;
    rc = ensure_slice_scratch(h, n_bb ? n_bb : 1);   // (growing a buffer frees the old one: hipFree synchronises the device)
    if (rc) return rc;
    RecvCall rcall = recv_call(h);
    rcall.reslice = h->track == P25FE_CLOCK_TRACKING_RESLICE;      // the whole range is in memory: SPEC 3.8c applies
    h->rs_n = 0;
    if (n_bb) {
        prof_begin(h);
        prof_mark(h, 0, st);
        hipEvent_t e0, e1;
        prof_k1_events(h, &e0, &e1);
        if (h->ext_events) {
            if (e1) k1_done = e1;                    // a sampled call: the profiling stop event doubles as "K1 done"
            else e1 = k1_done;
            k1_done_attached = true;
        }
        rc = launch_frontend(h, d_iq, fmt, ch_stride, 0, n, 0, -(long)PLPAD - h->look, nullptr, 0, nullptr, st, &g, 0, e0, e1);
        if (rc) return rc;
        prof_mark(h, 1, st);
    }
    if (!k1_done_attached) HIPCHK(h, hipEventRecord(k1_done, st));
    HIPCHK(h, hipStreamWaitEvent(h->rx_stream, k1_done, 0));
    if (n_bb) {
        rc = launch_detect(h, n_bb, -h->look, h->rx_stream, rcall, false);
        if (rc) return rc;
        prof_mark(h, 2, h->rx_stream);
    }
    rc = launch_scan_slice(h, n_bb, -h->look, nullptr, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result, n_bb != 0, h->rx_stream,
                           rcall, n_bb != 0, h->ext_events ? h->ev_rx[lane] : nullptr);
    h->prof_slot = -1;
    if (rc) return rc;
    if (!h->ext_events) HIPCHK(h, hipEventRecord(h->ev_rx[lane], h->rx_stream));
    h->rx_pending[lane] = true;
    return P25FE_OK;
}

int p25fe_run_dev(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n, uint8_t* d_dibits,
                  size_t dibit_stride, p25fe_result_t* d_result, void* stream)
{
    if (!h || !d_iq || !d_dibits || !d_result || p25fe_n_baseband_h(h, 0, n) > MAX_RANGE_BB) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    shard_invalidate(h);
    if (int jrc = pipe_join(h, st)) return jrc;
    const size_t n_bb = p25fe_n_baseband_h(h, 0, n);
    int rc = ensure_slice_scratch(h, n_bb ? n_bb : 1);
    if (rc) return rc;
    RecvCall rcall = recv_call(h);
    rcall.reslice = h->track == P25FE_CLOCK_TRACKING_RESLICE;      // the whole range is in memory: SPEC 3.8c applies
    h->rs_n = 0;
    if (n_bb == 0)
        return launch_scan_slice(h, 0, -h->look, nullptr, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result, false, st, rcall, false);
    const PlanarGeo g(n_bb);
    prof_begin(h);
    prof_mark(h, 0, st);
    hipEvent_t e0, e1;
    prof_k1_events(h, &e0, &e1);
    // K1 writes the baseband straight into the polyphase layout (+ sign planes); the 240 history positions in front
    // of the stream come out as the zeros of a fresh DemodTask (outputs of an all-zero input)
    rc = launch_frontend(h, d_iq, fmt, ch_stride, 0, n, 0, -(long)PLPAD - h->look, nullptr, 0, nullptr, st, &g, 0, e0, e1);
    if (rc) return rc;
    prof_mark(h, 1, st);
    rc = launch_detect(h, n_bb, -h->look, st, rcall, false);
    if (rc) return rc;
    prof_mark(h, 2, st);
    rc = launch_scan_slice(h, n_bb, -h->look, nullptr, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result, true, st, rcall, true);
    h->prof_slot = -1;
    return rc;
}

// --------------------------------------------------------------------------------------------
// time shards
// --------------------------------------------------------------------------------------------
// what: bit 0 = the main K1 launch, bit 1 = the head segment(s), bit 2 = sync detection + scan
enum { SH_MAIN = 1, SH_HEAD = 2, SH_RECV = 4 };
static int shard_pass1_part(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                            p25fe_result_t* d_result, hipStream_t st, int what)
{
    const bool do_main = (what & SH_MAIN) != 0, do_finish = (what & SH_RECV) != 0;
    bool do_head = (what & SH_HEAD) != 0;
    if (!h || !d_iq || (do_finish && !d_result) || p25fe_n_baseband_h(h, abs0, n) > MAX_RANGE_BB || piecewise_refused(h)) return P25FE_ERR_ARG;
    if (do_main) shard_invalidate(h);
    if (n_hist < SHARD_HALO && n_hist != abs0) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->sh_pipe)
        if (int jrc = pipe_join(h, st)) return jrc;
    const size_t n_bb = p25fe_n_baseband_h(h, abs0, n);
    const long abs_bb0 = (long)p25fe_n_baseband_h(h, 0, (size_t)abs0) - h->look;      // first processed baseband index of this shard
    int rc = ensure_slice_scratch(h, n_bb ? n_bb : 1);
    if (rc) return rc;
    const PlanarGeo g(n_bb);
    // the receiver's 240 history samples are recomputed from the IQ halo (zeros before the start of the stream)
    if (do_main) {
        prof_begin(h);
        prof_mark(h, 0, st);
        hipEvent_t k1_done = h->sh_pipe ? h->ev_k1[h->lane] : nullptr;
        bool k1_done_attached = false;
        if (n_bb) {
            // K1's event pair rides on THIS launch: with the split form it times the main launch alone (the head segment
            // that follows the halo wait is one workgroup), so an RCCL wait between the two is not in K1's figure
            hipEvent_t e0, e1;
            prof_k1_events(h, &e0, &e1);
            if (h->sh_pipe && h->ext_events) {
                if (e1) k1_done = e1;                               // a sampled call: the profiling stop event doubles as "K1 done"
                else e1 = k1_done;
                k1_done_attached = true;
            }
            rc = launch_frontend(h, d_iq, fmt, ch_stride, n_hist, n, abs0, -(long)PLPAD - h->look, nullptr, 0, nullptr, st, &g,
                                 do_head ? 0 : 1, e0, e1);
            if (rc) return rc;
        }
        if (h->sh_pipe) {
            // pipelined step: everything behind this launch runs on the receive stream, which waits for it here
            // (K1 on a CU-masked stream of its own, to keep a few CUs free for the exchanges that run beside it, was tried: the
            // masked launch itself ran 30 - 60 % slower -- profiles/r05_shard_pipelined.txt)
            if (!k1_done_attached) HIPCHK(h, hipEventRecord(k1_done, st));
            HIPCHK(h, hipStreamWaitEvent(h->rx_stream, k1_done, 0));
        }
        h->sh_main_nbb = n_bb; h->sh_main_abs0 = abs0;
        h->sh_head_done = do_head;
        do_head = false;                                            // (part 0 = every segment)
    } else {
        if (h->sh_main_nbb != n_bb || h->sh_main_abs0 != abs0) return P25FE_ERR_ARG;     // head / finish without its main launch
    }
    if ((do_head || do_finish) && !h->sh_head_done) {
        // the segments whose input reaches into the halo.  They share no byte of the planes with the main launch's
        // (SEG_HALO in p25fe_kernels.hip), so this launch may run BESIDE it on another stream (p25fe_shard_pass1_head).
        long head_end = 0;                                          // planar positions [0, head_end) are the head's
        const bool own_stream = !do_finish;                         // launched on its own (another stream's)
        h->sh_head_flagged = false;
        if (n_bb) {
            if (own_stream) {
                // the launch's last workgroup publishes "the head is in memory"; the detection's first tiles wait for that
                // word instead of the whole stream waiting for an event
                if (!h->sh_flag.p) {
                    HIPCHK(h, h->sh_flag.ensure(SH_FLAG_WORDS * sizeof(unsigned)));
                    HIPCHK(h, hipMemsetAsync(h->sh_flag.p, 0, SH_FLAG_WORDS * sizeof(unsigned), st));
                }
                ++h->sh_seq;
                if (h->sh_seq == 0u) ++h->sh_seq;
            }
            rc = launch_frontend(h, d_iq, fmt, ch_stride, n_hist, n, abs0, -(long)PLPAD - h->look, nullptr, 0, nullptr, st, &g, 2, nullptr, nullptr,
                                 nullptr, &head_end, own_stream ? h->sh_flag.as<unsigned>() : nullptr, h->sh_seq);
            if (rc) return rc;
            if (own_stream && head_end > 0) {                       // (no head workgroup, no flag: nothing to wait for)
                h->sh_head_flagged = true;
                h->sh_head_tile_max = (int)(head_end / TS);         // a tile reads planes up to its own end (+ one sign word): tiles past the head's never touch it
            }
        }
        h->sh_head_done = true;
    }
    if (!do_finish) return P25FE_OK;
    prof_mark(h, 1, st);
    const RecvCall rcall = recv_call(h);
    h->rs_n = 0;
    if (n_bb) {
        rc = launch_detect(h, n_bb, abs_bb0, st, rcall, h->sh_head_flagged);
        if (rc) return rc;
    }
    h->sh_head_flagged = false;
    prof_mark(h, 2, st);
    rc = launch_scan_slice(h, n_bb, abs_bb0, nullptr, nullptr, 0, nullptr, nullptr, 0, d_result, false, st, rcall, n_bb != 0);
    h->prof_slot = -1;
    if (rc) return rc;
    h->sh_valid = true; h->sh_nbb = n_bb; h->sh_abs_bb0 = abs_bb0; h->sh_gen = rcall.gen; h->sh_scan_fresh = true;
    h->sh_main_nbb = (size_t)-1; h->sh_head_done = false;
    return P25FE_OK;
}

int p25fe_shard_pass1(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                      p25fe_result_t* d_result, void* stream)
{
    return shard_pass1_part(h, d_iq, fmt, ch_stride, n_hist, n, abs0, d_result, (hipStream_t)stream, SH_MAIN | SH_HEAD | SH_RECV);
}

int p25fe_shard_pass1_main(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                           void* stream)
{
    return shard_pass1_part(h, d_iq, fmt, ch_stride, n_hist, n, abs0, nullptr, (hipStream_t)stream, SH_MAIN);
}

int p25fe_shard_pass1_head(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                           void* stream)
{
    return shard_pass1_part(h, d_iq, fmt, ch_stride, n_hist, n, abs0, nullptr, (hipStream_t)stream, SH_HEAD);
}

int p25fe_shard_pass1_k1(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0, void* stream)
{
    return shard_pass1_part(h, d_iq, fmt, ch_stride, n_hist, n, abs0, nullptr, (hipStream_t)stream, SH_MAIN | SH_HEAD);
}

int p25fe_shard_pass1_finish(p25fe_t* h, const void* d_iq, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                             p25fe_result_t* d_result, void* stream)
{
    return shard_pass1_part(h, d_iq, fmt, ch_stride, n_hist, n, abs0, d_result, (hipStream_t)stream, SH_RECV);
}

// The pipelined step of p25fe_rccl.cpp (p25fe_shard_step_pipelined): between _begin and _end the shard passes use the scratch set
// p25fe_run_dev_pipelined would use next; p25fe_shard_pass1_main stays on the caller's stream and makes the receive stream wait for
// it, every later pass is given the receive stream.  _end marks that stream's work as pending (p25fe_join_dev waits for it).
int p25fe_shard_pipe_begin(p25fe_t* h, void* stream, void** rx_stream)
{
    if (!h || !rx_stream || h->sh_pipe) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    shard_invalidate(h);
    if (int orc = pipe_open(h, (hipStream_t)stream, h->sh_depth)) return orc;
    h->sh_pipe = true;
    *rx_stream = h->rx_stream;
    return P25FE_OK;
}

int p25fe_rx_stream(p25fe_t* h, void** rx_stream)
{
    if (!h || !rx_stream) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rc = ensure_rx_stream(h)) return rc;
    *rx_stream = h->rx_stream;
    return P25FE_OK;
}

int p25fe_shard_pipe_end(p25fe_t* h, void* last_stream)
{
    if (!h || !h->sh_pipe) return P25FE_ERR_ARG;
    h->sh_pipe = false;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipEventRecord(h->ev_rx[h->lane], last_stream ? (hipStream_t)last_stream : h->rx_stream));
    h->rx_pending[h->lane] = true;
    return P25FE_OK;
}

int p25fe_shard_pass2(p25fe_t* h, const p25fe_anchor_t* d_anchor_in, uint8_t* d_dibits, size_t dibit_stride,
                      p25fe_result_t* d_result, void* stream)
{
    if (!h || !d_dibits || !d_result || !h->sh_valid) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    prof_begin(h);
    prof_mark(h, 2, (hipStream_t)stream);
    RecvCall rcall = recv_call(h);
    rcall.gen = h->sh_gen;                               // K3 / K4 read what pass 1's K2 left (the lock drops are in its summaries)
    h->sh_scan_fresh = false;                            // the re-scan below rewrites the groups' carry-ins with THIS carry-in
    const int rc = launch_scan_slice(h, h->sh_nbb, h->sh_abs_bb0, d_anchor_in, d_dibits, dibit_stride, nullptr, nullptr,
                                     0, d_result, true, (hipStream_t)stream, rcall, false);
    h->prof_slot = -1;
    return rc;
}

int p25fe_shard_pass2_dev(p25fe_t* h, const p25fe_result_t* d_summaries, const uint64_t* d_shard_bb0, const uint64_t* d_shard_bb_n,
                          size_t n_shards, size_t rank, p25fe_anchor_t* d_anchor_in, uint64_t* d_dibit_offset, uint8_t* d_dibits,
                          size_t dibit_stride, uint8_t* d_dibits_dup, p25fe_result_t* d_result, void* stream)
{
    if (!h || !d_summaries || !d_shard_bb0 || !d_shard_bb_n || !d_anchor_in || !d_dibit_offset || !d_dibits || !d_result || !h->sh_valid ||
        n_shards == 0 || n_shards > 0x7fffffffu || rank >= n_shards || h->C != 1)
        return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    RecvCall rcall = recv_call(h);
    rcall.gen = h->sh_gen;
    if (rcall.gen || !h->sh_scan_fresh) {
        // tracking clock / lock drops inside the shard (or a p25fe_shard_pass2 has already rewritten pass 1's scan): the combine
        // as its own (one-thread) launch, then the re-scan of the shard's groups + slicer with that carry-in
        h->sh_scan_fresh = false;
        hipLaunchKernelGGL(k_shard_resolve, dim3(1), dim3(64), 0, st, d_summaries, d_shard_bb0, d_shard_bb_n, (int)n_shards, h->track,
                           d_anchor_in, d_dibit_offset);
        HIPCHK(h, hipGetLastError());
        prof_begin(h);
        prof_mark(h, 2, st);
        const int rc = launch_scan_slice(h, h->sh_nbb, h->sh_abs_bb0, d_anchor_in + rank, d_dibits, dibit_stride, nullptr, nullptr, 0,
                                         d_result, true, st, rcall, false, nullptr, d_dibits_dup);
        h->prof_slot = -1;
        return rc;
    }
    ShardFix fx;
    fx.summ = d_summaries; fx.bb0 = d_shard_bb0; fx.bbn = d_shard_bb_n; fx.n_shards = (int)n_shards; fx.rank = (int)rank;
    fx.anc_out = d_anchor_in; fx.off_out = d_dibit_offset; fx.result = d_result;
    prof_begin(h);
    prof_mark(h, 2, st);
    int rc;
    if (h->sh_nbb == 0) {
        // a shard without a baseband sample has no slicer tile to run the combine in: separate launches (never the hot path)
        hipLaunchKernelGGL(k_shard_resolve, dim3(1), dim3(64), 0, st, d_summaries, d_shard_bb0, d_shard_bb_n, (int)n_shards, h->track,
                           d_anchor_in, d_dibit_offset);
        HIPCHK(h, hipGetLastError());
        rc = launch_scan_slice(h, 0, h->sh_abs_bb0, d_anchor_in + rank, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result, true, st, rcall, false);
    } else {
        rc = launch_scan_slice(h, h->sh_nbb, h->sh_abs_bb0, nullptr, d_dibits, dibit_stride, nullptr, nullptr, 0, d_result, true, st, rcall,
                               true, nullptr, d_dibits_dup, &fx);
    }
    h->prof_slot = -1;
    return rc;
}

int p25fe_shard_resolve(const p25fe_result_t* summaries, const uint64_t* shard_bb0, const uint64_t* shard_bb_n,
                        size_t n_shards, int symbol_clock, p25fe_anchor_t* anchor_in, uint64_t* dibit_offset)
{
    if (!summaries || !shard_bb0 || !shard_bb_n || !anchor_in || !dibit_offset || n_shards > 0x7fffffffu) return P25FE_ERR_ARG;
    shard_resolve_impl(summaries, shard_bb0, shard_bb_n, (int)n_shards, symbol_clock, anchor_in, dibit_offset);
    return P25FE_OK;
}

int p25fe_shard_resolve_dev(p25fe_t* h, const p25fe_result_t* d_summaries, const uint64_t* d_shard_bb0,
                            const uint64_t* d_shard_bb_n, size_t n_shards, p25fe_anchor_t* d_anchor_in,
                            uint64_t* d_dibit_offset, void* stream)
{
    if (!h || !d_summaries || !d_shard_bb0 || !d_shard_bb_n || !d_anchor_in || !d_dibit_offset || n_shards > 0x7fffffffu) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_shard_resolve, dim3(1), dim3(64), 0, (hipStream_t)stream, d_summaries, d_shard_bb0, d_shard_bb_n,
                       (int)n_shards, h->track, d_anchor_in, d_dibit_offset);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_shard_head_check(p25fe_t* h)
{
    if (!h) return P25FE_ERR_ARG;
    if (!h->sh_flag.p) return P25FE_OK;                              // no head segment has ever run on a stream of its own
    HIPCHK(h, hipSetDevice(h->cfg.device));
    unsigned err = 0u;
    HIPCHK(h, hipMemcpy(&err, h->sh_flag.as<unsigned>() + SH_FLAG_ERR, sizeof err, hipMemcpyDeviceToHost));
    return err ? P25FE_ERR_TIMEOUT : P25FE_OK;
}

int p25fe_streams_share_queue(p25fe_t* h, void* stream_a, void* stream_b, int* shared)
{
    if (!h || !shared) return P25FE_ERR_ARG;
    *shared = 0;
    if (stream_a == stream_b) { *shared = 1; return P25FE_OK; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->sh_flag.p) {
        HIPCHK(h, h->sh_flag.ensure(SH_FLAG_WORDS * sizeof(unsigned)));
        HIPCHK(h, hipMemset(h->sh_flag.p, 0, SH_FLAG_WORDS * sizeof(unsigned)));
    }
    unsigned* w = h->sh_flag.as<unsigned>() + 2;                     // (words 0 / 1 are the head segment's flag and ticket)
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream_a));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream_b));
    int votes = 0;
    for (int rep = 0; rep < 3; ++rep) {                              // (three rounds: a busy chip can delay the setter once)
        HIPCHK(h, hipMemset(w, 0, 2 * sizeof(unsigned)));
        hipLaunchKernelGGL(k_queue_probe_wait, dim3(1), dim3(1), 0, (hipStream_t)stream_a, w, 20000ull);      // 200 us at 100 MHz
        hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(1), 0, (hipStream_t)stream_b, w);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize((hipStream_t)stream_a));
        HIPCHK(h, hipStreamSynchronize((hipStream_t)stream_b));
        unsigned r[2] = {0u, 0u};
        HIPCHK(h, hipMemcpy(r, w, sizeof r, hipMemcpyDeviceToHost));
        if (r[1] == 2u) ++votes;
    }
    *shared = votes == 3;
    return P25FE_OK;
}

int p25fe_shard_compact_dev(p25fe_t* h, const uint8_t* d_gathered, size_t cap, const uint64_t* d_dibit_offset,
                            size_t n_shards, uint8_t* d_out, size_t out_cap, void* stream)
{
    if (!h || !d_gathered || !d_dibit_offset || !d_out || n_shards == 0 || n_shards > 65535) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const unsigned bx = (unsigned)((cap + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(k_shard_compact, dim3(bx ? bx : 1, (unsigned)n_shards), dim3(256), 0, (hipStream_t)stream, d_gathered,
                       (unsigned long long)cap, d_dibit_offset, 0, (int)n_shards, d_out, (unsigned long long)out_cap);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_shard_compact_from_dev(p25fe_t* h, const uint8_t* d_gathered, size_t cap, const uint64_t* d_dibit_offset, size_t first_shard,
                                 size_t n_shards, uint8_t* d_out, size_t out_cap, void* stream)
{
    if (!h || !d_gathered || !d_dibit_offset || !d_out || n_shards == 0 || n_shards > 65535 || first_shard > n_shards) return P25FE_ERR_ARG;
    if (first_shard == n_shards) return P25FE_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const unsigned bx = (unsigned)((cap + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(k_shard_compact, dim3(bx ? bx : 1, (unsigned)(n_shards - first_shard)), dim3(256), 0, (hipStream_t)stream, d_gathered,
                       (unsigned long long)cap, d_dibit_offset, (int)first_shard, (int)n_shards, d_out, (unsigned long long)out_cap);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

// --------------------------------------------------------------------------------------------
// streaming with host buffers: the bodies of DemodTask::run / RecvTask::run, one chunk per call
//
// All stream state (IQ history, baseband tail, anchors, counters) lives in HOST memory; input is staged in pinned,
// device-visible memory as [history | new samples] and read by the kernels over PCIe (zero copy: a 32 768-byte chunk is
// one latency, not a DMA command); results are written by the kernels straight into pinned memory.  A call is
// launches + ONE synchronisation, and for a chunk of at most one tile (the reference's 16 384-sample buffer gives
// 3 276 / 3 277 baseband samples) ONE launch: k_chunk / k_recv_chunk.
// --------------------------------------------------------------------------------------------
// [history | new] per channel in h->hin; returns the staging geometry
struct Staged { size_t stride, n_hist; char* host; const char* dev; };
static int stage_iq(p25fe_t* h, const void* iq, int fmt, size_t n, Staged* s)
{
    if (h->fmt_locked >= 0 && h->fmt_locked != fmt && h->abs_iq > 0) return P25FE_ERR_FORMAT;
    const size_t C = (size_t)h->C, eb = fmt_bytes(fmt);
    s->stride = SHARD_HALO + round_up(n, 8) + 8;                     // samples, multiple of 8: owned sample 0 stays 16-B aligned
    HIPCHK(h, h->hin.ensure(C * s->stride * eb));
    s->host = static_cast<char*>(h->hin.p);
    s->dev = static_cast<const char*>(h->hin.dp);
    s->n_hist = h->abs_iq < SHARD_HALO ? (size_t)h->abs_iq : SHARD_HALO;
    for (size_t c = 0; c < C; ++c) {
        char* row = s->host + c * s->stride * eb;
        memcpy(row, h->hist_iq.data() + c * SHARD_HALO * 8, SHARD_HALO * eb);
        if (n) memcpy(row + SHARD_HALO * eb, static_cast<const char*>(iq) + c * n * eb, n * eb);
    }
    return P25FE_OK;
}
// the call succeeded: the last SHARD_HALO samples of [history | new] become the history
static void commit_iq(p25fe_t* h, int fmt, size_t n, const Staged& s)
{
    const size_t C = (size_t)h->C, eb = fmt_bytes(fmt);
    for (size_t c = 0; c < C; ++c)
        memcpy(h->hist_iq.data() + c * SHARD_HALO * 8, s.host + (c * s.stride + n) * eb, SHARD_HALO * eb);
    h->fmt_locked = fmt;
    h->abs_iq += n;
}

static int demod_host(p25fe_t* h, const void* iq, int fmt, size_t n, float* bb, size_t bb_cap, size_t* n_out,
                      float* power_dbm)
{
    if (!h || (!iq && n) || !bb || !n_out) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t nb = p25fe_n_baseband_h(h, h->abs_iq, n);
    if (nb > bb_cap) return P25FE_ERR_CAPACITY;
    const size_t C = (size_t)h->C;
    Staged sg;
    int rc = stage_iq(h, iq, fmt, n, &sg);
    if (rc) return rc;
    const size_t bb_stride = round_up(nb + 4, 4);
    HIPCHK(h, h->hout.ensure(C * bb_stride * sizeof(float) + 64 * (C + 2)));
    Arena ar(h->hout);
    float *d_bb, *d_pw;
    float* h_bb = ar.take<float>(C * bb_stride, &d_bb);
    float* h_pw = ar.take<float>(C, &d_pw);
    rc = launch_frontend(h, sg.dev + SHARD_HALO * fmt_bytes(fmt), fmt, sg.stride, sg.n_hist, n, h->abs_iq, 0, d_bb, bb_stride,
                         power_dbm ? d_pw : nullptr, h->stream);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t c = 0; c < C; ++c) memcpy(bb + c * bb_cap, h_bb + c * bb_stride, nb * sizeof(float));
    if (power_dbm) memcpy(power_dbm, h_pw, sizeof(float) * C);
    commit_iq(h, fmt, n, sg);
    *n_out = nb;
    return P25FE_OK;
}

int p25fe_demod_u8(p25fe_t* h, const uint8_t* iq, size_t n_bytes, float* bb, size_t bb_cap, size_t* n_out,
                   float* power_dbm)
{
    if (n_bytes & 1) return P25FE_ERR_ARG;
    return demod_host(h, iq, P25FE_FMT_U8, n_bytes / 2, bb, bb_cap, n_out, power_dbm);
}

int p25fe_demod_cf32(p25fe_t* h, const float* iq, size_t n_samples, float* bb, size_t bb_cap, size_t* n_out,
                     float* power_dbm)
{
    return demod_host(h, iq, P25FE_FMT_CF32, n_samples, bb, bb_cap, n_out, power_dbm);
}

// Pinned outputs of one receiver call: per channel a result record, the carry-in anchor, a dibit row and (optionally) the
// sync event rows; plus the baseband tail the fused path hands back.
struct RecvOut {
    p25fe_result_t *res, *d_res;
    p25fe_anchor_t *anc, *d_anc;
    uint8_t *dib, *d_dib;
    int64_t *spos, *d_spos;
    uint64_t *sdib, *d_sdib;
    float *tail, *d_tail;
    unsigned *done, *d_done;             // completion words the chunk kernels write last (polled by the host)
    size_t dstride, sstride;
};
static int recv_out(p25fe_t* h, size_t n_bb, size_t sync_cap, RecvOut* o)
{
    const size_t C = (size_t)h->C;
    o->dstride = round_up(n_bb / (W + 1) + 2, 16);                   // hard ceiling: detections, hence re-anchors, are at least W + 1 samples apart
    o->sstride = sync_cap;
    const size_t bytes = C * (sizeof(p25fe_result_t) + sizeof(p25fe_anchor_t) + o->dstride + 16 * sync_cap + TAILN * sizeof(float) + 4) + 64 * 8;
    HIPCHK(h, h->hout.ensure(bytes));
    Arena ar(h->hout);
    o->res = ar.take<p25fe_result_t>(C, &o->d_res);
    o->anc = ar.take<p25fe_anchor_t>(C, &o->d_anc);
    o->dib = ar.take<uint8_t>(C * o->dstride, &o->d_dib);
    o->spos = ar.take<int64_t>(C * sync_cap, &o->d_spos);
    o->sdib = ar.take<uint64_t>(C * sync_cap, &o->d_sdib);
    o->tail = ar.take<float>(C * TAILN, &o->d_tail);
    o->done = ar.take<unsigned>(C, &o->d_done);
    memcpy(o->anc, h->anchor.data(), sizeof(p25fe_anchor_t) * C);
    return P25FE_OK;
}
// after the synchronisation: capacity check, copy-out, receiver state
static int recv_finish(p25fe_t* h, const RecvOut& o, size_t n_bb, uint8_t* dibits, size_t cap, size_t* n_dibits, int64_t* sync_pos,
                       uint64_t* sync_dibit, size_t sync_cap, size_t* n_sync)
{
    const size_t C = (size_t)h->C;
    for (size_t c = 0; c < C; ++c)
        if (o.res[c].n_dibits > cap) return P25FE_ERR_CAPACITY;      // before any state moves: the call can be repeated with more room
    for (size_t c = 0; c < C; ++c) {
        const p25fe_result_t& r = o.res[c];
        memcpy(dibits + c * cap, o.dib + c * o.dstride, (size_t)r.n_dibits);
        const size_t ns = r.n_sync < sync_cap ? (size_t)r.n_sync : sync_cap;
        if (ns && sync_pos) memcpy(sync_pos + c * sync_cap, o.spos + c * o.sstride, ns * sizeof(int64_t));
        if (sync_dibit)
            for (size_t k = 0; k < ns; ++k) sync_dibit[c * sync_cap + k] = o.sdib[c * o.sstride + k] + h->total_dibits[c];
        n_dibits[c] = (size_t)r.n_dibits;
        if (n_sync) n_sync[c] = (size_t)r.n_sync;
        h->anchor[c] = r.anchor_out;
        h->total_dibits[c] += r.n_dibits;
    }
    h->abs_bb += n_bb;
    return P25FE_OK;
}

static void chunk_recv_args(p25fe_t* h, const RecvOut& o, size_t n_bb, long view0, bool want_sync, bool want_tail, ChunkRecvArgs* c)
{
    const PlanarGeo g(n_bb);
    c->pl = planar_view(h, g); c->n = (long)n_bb; c->abs0 = view0;
    c->recs = h->recs.as<TileRec>(); c->tsum = h->tsum.as<unsigned long long>(); c->evl = h->evl.as<uint16_t>(); c->evthr = h->evthr.as<float>();
    c->anchor_in = o.d_anc; c->result = o.d_res; c->dibits = o.d_dib; c->dibit_stride = (long)o.dstride;
    c->sync_pos = want_sync ? o.d_spos : nullptr; c->sync_dibit = want_sync ? o.d_sdib : nullptr; c->sync_stride = (long)o.sstride;
    c->tail = want_tail ? o.d_tail : nullptr; c->look = (int)h->look; c->n_baseband = n_bb;
    c->done = nullptr; c->seq = 0u;
}

// One-launch chunk calls finish by POLLING a word the kernel writes last into pinned memory (a system-scope release in front
// of it) instead of hipStreamSynchronize: the completion signal -> interrupt -> wake-up path costs more than the kernel.
// P25FE_CHUNK_POLL=0 keeps the synchronisation.  The stream is synchronised anyway if the word does not arrive in time.
static void chunk_poll_arm(p25fe_t* h, const RecvOut& o, ChunkRecvArgs* c)
{
    static const bool poll = [] { const char* e = getenv("P25FE_CHUNK_POLL"); return !(e && atoi(e) == 0); }();
    if (!poll) return;
    ++h->chunk_seq;
    if (h->chunk_seq == 0u) ++h->chunk_seq;
    for (int ch = 0; ch < h->C; ++ch) o.done[ch] = 0u;
    c->done = o.d_done; c->seq = h->chunk_seq;
}
static int chunk_wait(p25fe_t* h, const RecvOut& o, const ChunkRecvArgs& c)
{
    if (c.done) {
        volatile unsigned* w = o.done;
        const size_t C = (size_t)h->C;
        for (long spin = 0; spin < 4000000L; ++spin) {               // ~ tens of milliseconds at worst, then fall back
            size_t k = 0;
            while (k < C && w[k] == c.seq) ++k;
            if (k == C) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return P25FE_OK; }
            __builtin_ia32_pause();
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return P25FE_OK;
}

int p25fe_slice(p25fe_t* h, const float* bb, size_t n, uint8_t* dibits, size_t cap, size_t* n_dibits, int64_t* sync_pos,
                uint64_t* sync_dibit, size_t sync_cap, size_t* n_sync)
{
    if (!h || (!bb && n) || !dibits || !n_dibits || piecewise_refused(h)) return P25FE_ERR_ARG;
    if ((sync_pos || sync_dibit) && !(sync_pos && sync_dibit)) return P25FE_ERR_ARG;
    if (!sync_pos) sync_cap = 0;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t C = (size_t)h->C;
    if (n == 0) { for (size_t c = 0; c < C; ++c) { n_dibits[c] = 0; if (n_sync) n_sync[c] = 0; } return P25FE_OK; }
    if (cap < n / SPS + 1) return P25FE_ERR_CAPACITY;               // worst case of an undisturbed lock, checked before any state moves
    shard_invalidate(h);
    if (int jrc = pipe_join(h, h->stream)) return jrc;
    // stage [tail | new] per channel
    const size_t bb_stride = round_up(BBPAD + n + 4, 4);
    HIPCHK(h, h->hbb.ensure(C * bb_stride * sizeof(float)));
    float* hb = static_cast<float*>(h->hbb.p);
    const float* db = static_cast<const float*>(h->hbb.dp);
    for (size_t c = 0; c < C; ++c) {
        memcpy(hb + c * bb_stride, h->tail_bb.data() + c * BBPAD, BBPAD * sizeof(float));
        memcpy(hb + c * bb_stride + BBPAD, bb + c * n, n * sizeof(float));
    }
    RecvOut o;
    int rc = recv_out(h, n, sync_cap, &o);
    if (rc) return rc;
    rc = ensure_slice_scratch(h, n);
    if (rc) return rc;
    ChunkRecvArgs polled;
    polled.done = nullptr; polled.seq = 0u;
    const size_t hist = h->abs_bb < BBPAD ? (size_t)h->abs_bb : BBPAD;
    const long view0 = (long)h->abs_bb - h->look;
    if (!h->track && !h->rs_n && n <= (size_t)TS) {
        RecvChunkArgs a;
        chunk_recv_args(h, o, n, view0, sync_cap != 0, false, &a.r);
        chunk_poll_arm(h, o, &a.r);
        polled = a.r;
        const PlanarGeo g(n);
        a.bb = db + BBPAD; a.bb_stride = (long)bb_stride; a.n_hist = (long)hist;
        a.f = h->pl_f.as<float>(); a.bits = h->pl_bits.as<uint32_t>(); a.n_blocks = (long)g.n_blocks;
        hipLaunchKernelGGL(k_recv_chunk, dim3((unsigned)C), dim3(WV), 0, h->stream, a);
        HIPCHK(h, hipGetLastError());
    } else {
        const RecvCall rcall = recv_call(h);                        // (a pending p25fe_resync_at_dev list belongs to this call: cleared below, once it has succeeded)
        rc = launch_planarize(h, db + BBPAD, bb_stride, hist, n, h->stream);
        if (rc) return rc;
        rc = launch_detect(h, n, view0, h->stream, rcall, false);
        if (rc) return rc;
        rc = launch_scan_slice(h, n, view0, o.d_anc, o.d_dib, o.dstride, sync_cap ? o.d_spos : nullptr,
                               sync_cap ? o.d_sdib : nullptr, o.sstride, o.d_res, true, h->stream, rcall, true);
        if (rc) return rc;
    }
    rc = chunk_wait(h, o, polled);
    if (rc) return rc;
    rc = recv_finish(h, o, n, dibits, cap, n_dibits, sync_pos, sync_dibit, sync_cap, n_sync);
    if (rc) return rc;                                               // (P25FE_ERR_CAPACITY: nothing has moved, the lock-drop list included -- repeat the call)
    h->rs_n = 0;
    for (size_t c = 0; c < C; ++c)                                   // the tail: last BBPAD samples of [tail | new]
        memcpy(h->tail_bb.data() + c * BBPAD, hb + c * bb_stride + n, BBPAD * sizeof(float));
    return P25FE_OK;
}

// launch of the fused chunk kernel: K1 (planar) + receiver tail
static int launch_chunk(p25fe_t* h, const void* d_x, int fmt, size_t ch_stride, size_t n_hist, size_t n, uint64_t abs0,
                        const ChunkRecvArgs& r)
{
    const PlanarGeo g(p25fe_n_baseband_h(h, abs0, n));
    return launch_frontend(h, d_x, fmt, ch_stride, n_hist, n, abs0, -(long)PLPAD - h->look, nullptr, 0, nullptr, h->stream, &g, 0,
                           nullptr, nullptr, &r);
}

static int run_host(p25fe_t* h, const void* iq, int fmt, size_t n, uint8_t* dibits, size_t cap, size_t* n_dibits)
{
    if (!h || (!iq && n) || !dibits || !n_dibits || piecewise_refused(h)) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t C = (size_t)h->C;
    const size_t nb = p25fe_n_baseband_h(h, h->abs_iq, n);
    if (cap < nb / SPS + 1) return P25FE_ERR_CAPACITY;              // worst case of an undisturbed lock, checked before any state moves
    shard_invalidate(h);
    if (int jrc = pipe_join(h, h->stream)) return jrc;
    Staged sg;
    int rc = stage_iq(h, iq, fmt, n, &sg);
    if (rc) return rc;
    if (nb == 0) {                                                  // fewer than five new samples: only the filters' history moves
        commit_iq(h, fmt, n, sg);
        for (size_t c = 0; c < C; ++c) n_dibits[c] = 0;
        return P25FE_OK;
    }
    RecvOut o;
    rc = recv_out(h, nb, 0, &o);
    if (rc) return rc;
    rc = ensure_slice_scratch(h, nb);
    if (rc) return rc;
    const size_t eb = fmt_bytes(fmt);
    const long view0 = (long)h->abs_bb - h->look;
    const PlanarGeo g(nb);
    ChunkRecvArgs cr;
    chunk_recv_args(h, o, nb, view0, false, true, &cr);
    bool one_launch = false;
    if (!h->track && !h->rs_n && nb <= (size_t)TS) {
        chunk_poll_arm(h, o, &cr);
        rc = launch_chunk(h, sg.dev + SHARD_HALO * eb, fmt, sg.stride, sg.n_hist, n, h->abs_iq, cr);
        if (rc) return rc;
        one_launch = true;
    } else {
        // the baseband stays in HBM; the receiver's history is recomputed from the IQ history, like a shard's from its halo
        const RecvCall rcall = recv_call(h);
        rc = launch_frontend(h, sg.dev + SHARD_HALO * eb, fmt, sg.stride, sg.n_hist, n, h->abs_iq, -(long)PLPAD - h->look, nullptr, 0,
                             nullptr, h->stream, &g);
        if (rc) return rc;
        rc = launch_detect(h, nb, view0, h->stream, rcall, false);
        if (rc) return rc;
        rc = launch_scan_slice(h, nb, view0, o.d_anc, o.d_dib, o.dstride, nullptr, nullptr, 0, o.d_res, true, h->stream, rcall, true);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tail_extract, dim3((unsigned)C), dim3(WV), 0, h->stream, cr);
        HIPCHK(h, hipGetLastError());
    }
    if (!one_launch) cr.done = nullptr;
    rc = chunk_wait(h, o, cr);
    if (rc) return rc;
    rc = recv_finish(h, o, nb, dibits, cap, n_dibits, nullptr, nullptr, 0, nullptr);
    if (rc) return rc;
    h->rs_n = 0;                                                     // the lock-drop list is consumed by a call that succeeded
    commit_iq(h, fmt, n, sg);
    // baseband tail for a later p25fe_slice on this handle: the newest TAILN samples, those older than what the planes hold
    // (a very short chunk) come from the previous tail
    for (size_t c = 0; c < C; ++c) {
        float* t = h->tail_bb.data() + c * BBPAD;
        float merged[BBPAD];
        for (size_t j = 0; j < (size_t)BBPAD; ++j) {
            const long m = (long)nb - (long)BBPAD + (long)j;        // index relative to this call's first baseband sample
            merged[j] = m >= -(long)(HIST_BB + h->look) ? o.tail[c * TAILN + j] : t[j + nb];
        }
        memcpy(t, merged, sizeof merged);
    }
    return P25FE_OK;
}

int p25fe_run_u8(p25fe_t* h, const uint8_t* iq, size_t n_bytes, uint8_t* dibits, size_t cap, size_t* n_dibits)
{
    if (n_bytes & 1) return P25FE_ERR_ARG;
    return run_host(h, iq, P25FE_FMT_U8, n_bytes / 2, dibits, cap, n_dibits);
}

int p25fe_run_cf32(p25fe_t* h, const float* iq, size_t n_samples, uint8_t* dibits, size_t cap, size_t* n_dibits)
{
    return run_host(h, iq, P25FE_FMT_CF32, n_samples, dibits, cap, n_dibits);
}

// --------------------------------------------------------------------------------------------
// a long host capture as a pipeline of windows: H2D copy | K1..K4 | dibits D2H, each on its own stream
// --------------------------------------------------------------------------------------------
int p25fe_run_host_windows(p25fe_t* h, const void* iq, int fmt, size_t n, size_t window, uint8_t* dibits, size_t cap,
                           size_t* n_dibits, p25fe_windows_stats_t* stats)
{
    if (!h || (!iq && n) || !dibits || !n_dibits || (fmt != P25FE_FMT_CF32 && fmt != P25FE_FMT_U8) || piecewise_refused(h)) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->fmt_locked >= 0 && h->fmt_locked != fmt && h->abs_iq > 0) return P25FE_ERR_FORMAT;
    const size_t C = (size_t)h->C, eb = fmt_bytes(fmt);
    if (window == 0) window = (size_t)(64u << 20) / eb;
    window &= ~(size_t)7;                                           // every window's first sample stays 16-byte aligned
    if (window < 8192) window = 8192;
    const size_t nb_total = p25fe_n_baseband_h(h, h->abs_iq, n);
    if (cap < nb_total / SPS + 1) return P25FE_ERR_CAPACITY;        // worst case of an undisturbed lock, before any state moves
    if (stats) memset(stats, 0, sizeof *stats);
    for (size_t c = 0; c < C; ++c) n_dibits[c] = 0;
    if (n == 0) return P25FE_OK;
    if (n <= window && n <= (size_t)(1u << 20)) {                   // one small window: the streaming call does the same with less set-up
        const int rc1 = run_host(h, iq, fmt, n, dibits, cap, n_dibits);
        if (stats) { stats->n_windows = 1; }
        return rc1;
    }
    shard_invalidate(h);
    if (int jrc = pipe_join(h, h->stream)) return jrc;
    const auto t_begin = std::chrono::steady_clock::now();
    constexpr int R = 4;
    if (!h->win_cs) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->win_cs, hipStreamNonBlocking));
        HIPCHK(h, hipStreamCreateWithFlags(&h->win_os, hipStreamNonBlocking));
        for (auto& slot : h->win_ev) for (hipEvent_t& e : slot) HIPCHK(h, hipEventCreate(&e));
    }
    hipStream_t st = h->stream, cs = h->win_cs, os = h->win_os;
    // A remainder of fewer than 8 samples (it may not even yield a baseband sample) rides with the window in front of it -- every
    // window of the pipeline then produces baseband, and the tail handed to a later p25fe_slice is always the last window's.
    size_t n_win = (n + window - 1) / window;
    const size_t rem = n - (n_win - 1) * window;                    // samples of the last window, 1 .. window
    const bool fold = n_win >= 2 && rem < 8;
    if (fold) --n_win;
    // (the last window is then `window + rem` samples long: the + 8 of the strides below)
    const size_t stride = SHARD_HALO + window + 8;                  // samples per channel row of a device window: [halo | window (+ < 8)]
    const size_t nb_win = (window + 8) / DEC + 2;
    const size_t dstride = round_up(nb_win / (W + 1) + 2, 64);      // hard ceiling of a window's dibits (p25fe_slice's rule)
    int rc = ensure_slice_scratch(h, nb_win);
    if (rc) return rc;
    for (int b = 0; b < 2; ++b) {
        HIPCHK(h, h->win_buf[b].ensure(C * stride * eb));
        HIPCHK(h, h->win_dib[b].ensure(C * dstride));
    }
    HIPCHK(h, h->win_res.ensure((size_t)R * C * sizeof(p25fe_result_t)));
    HIPCHK(h, h->win_anc.ensure(2 * C * sizeof(p25fe_anchor_t)));
    // pinned: per ring slot the result records, per parity a dibit row block, the baseband tail, the carry-in anchors, the history
    const size_t out_bytes = (size_t)R * round_up(C * sizeof(p25fe_result_t), 64) + 2 * round_up(C * dstride, 64) +
                             round_up(C * TAILN * sizeof(float), 64) + round_up(C * sizeof(p25fe_anchor_t), 64) + round_up(C * SHARD_HALO * eb, 64);
    HIPCHK(h, h->win_out.ensure(out_bytes));
    Arena ar(h->win_out);
    p25fe_result_t *h_res[R], *dv_unused_r;
    for (int q = 0; q < R; ++q) h_res[q] = ar.take<p25fe_result_t>(C, &dv_unused_r);
    uint8_t *h_dib[2], *dv_unused_b;
    for (int b = 0; b < 2; ++b) h_dib[b] = ar.take<uint8_t>(C * dstride, &dv_unused_b);
    float *d_tail, *h_tail = ar.take<float>(C * TAILN, &d_tail);
    p25fe_anchor_t *dv_unused_a, *h_anc = ar.take<p25fe_anchor_t>(C, &dv_unused_a);
    char *dv_unused_h, *h_hist = ar.take<char>(C * SHARD_HALO * eb, &dv_unused_h);
    // is the caller's memory pinned?  (then the copy engine reads it directly; otherwise this thread stages it)
    bool pinned = false;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, iq) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    if (!pinned)
        for (int b = 0; b < 2; ++b) HIPCHK(h, h->win_stage[b].ensure(C * (SHARD_HALO + window + 8) * eb));
    memcpy(h_anc, h->anchor.data(), C * sizeof(p25fe_anchor_t));
    for (size_t c = 0; c < C; ++c) memcpy(h_hist + c * SHARD_HALO * eb, h->hist_iq.data() + c * SHARD_HALO * 8, SHARD_HALO * eb);
    p25fe_anchor_t* d_anc = h->win_anc.as<p25fe_anchor_t>();        // [2][C]: the carry-in of window k lives in half k & 1
    const RecvCall rcall = recv_call(h);                             // (a pending lock-drop list holds absolute indices: every window sees it)
    std::vector<uint64_t> total(C, 0);
    double ms_h2d = 0.0, ms_comp = 0.0;
    int status = P25FE_OK;
    auto drain = [&](size_t j) -> int {                              // window j's dibits are on the host: append them
        hipEvent_t* ev = h->win_ev[j % R];
        HIPCHK(h, hipEventSynchronize(ev[4]));
        float t = 0.f;
        if (hipEventElapsedTime(&t, ev[0], ev[1]) == hipSuccess) ms_h2d += t; else (void)hipGetLastError();
        if (hipEventElapsedTime(&t, ev[2], ev[3]) == hipSuccess) ms_comp += t; else (void)hipGetLastError();
        const p25fe_result_t* r = h_res[j % R];
        for (size_t c = 0; c < C; ++c) {
            const uint64_t nd = r[c].n_dibits;
            if (nd > dstride || total[c] + nd > cap) return P25FE_ERR_CAPACITY;
            memcpy(dibits + c * cap + total[c], h_dib[j & 1] + c * dstride, (size_t)nd);
            total[c] += nd;
        }
        return P25FE_OK;
    };
    size_t nb_done = 0;
// (inside the window loop an error must not return at once: copies that read the caller's capture are in flight)
#define WINCHK(expr) if (const hipError_t e__ = (expr); e__ != hipSuccess) { h->last_hip = (int)e__; status = P25FE_ERR_HIP; break; } else (void)0
    for (size_t k = 0; k < n_win && status == P25FE_OK; ++k) {
        const size_t off = k * window, wn = k + 1 == n_win ? n - off : window;
        const uint64_t abs0 = h->abs_iq + off;
        const size_t n_hist = k == 0 ? (h->abs_iq < SHARD_HALO ? (size_t)h->abs_iq : SHARD_HALO) : SHARD_HALO;
        const int b = (int)(k & 1);
        hipEvent_t* ev = h->win_ev[k % R];
        char* dev = h->win_buf[b].as<char>();
        // ---- copy stream: [halo | window] in ONE copy.  The halo of window k >= 1 is simply the SHARD_HALO samples in front of
        // it in the caller's capture (16 KB more per window); only window 0 takes it from the handle's history.  (A device-to-
        // device copy of the previous window's tail put a second operation between every two H2D copies: 7 - 9 % of the call.)
        if (k >= 2) { WINCHK(hipStreamWaitEvent(cs, h->win_ev[(k - 2) % R][3], 0)); }      // the kernels of window k - 2 have read this buffer
        const size_t lead = k == 0 ? 0 : SHARD_HALO;                // samples in front of the window that travel with it
        if (k == 0) { WINCHK(hipMemcpy2DAsync(dev, stride * eb, h_hist, SHARD_HALO * eb, SHARD_HALO * eb, C, hipMemcpyHostToDevice, cs)); }
        const char* src = static_cast<const char*>(iq) + (off - lead) * eb;
        size_t spitch = n * eb;
        if (!pinned) {
            if (k >= 2) { WINCHK(hipEventSynchronize(h->win_ev[(k - 2) % R][1])); }       // the copy engine is done with this staging window
            char* sg = static_cast<char*>(h->win_stage[b].p);
            for (size_t c = 0; c < C; ++c) memcpy(sg + c * (lead + wn) * eb, src + c * n * eb, (lead + wn) * eb);
            src = sg; spitch = (lead + wn) * eb;
        }
        char* dst = dev + (SHARD_HALO - lead) * eb;
        WINCHK(hipEventRecord(ev[0], cs));
        if (C == 1) { WINCHK(hipMemcpyAsync(dst, src, (lead + wn) * eb, hipMemcpyHostToDevice, cs)); }
        else { WINCHK(hipMemcpy2DAsync(dst, stride * eb, src, spitch, (lead + wn) * eb, C, hipMemcpyHostToDevice, cs)); }
        WINCHK(hipEventRecord(ev[1], cs));
        // ---- compute stream
        WINCHK(hipStreamWaitEvent(st, ev[1], 0));
        if (k >= 2) { WINCHK(hipStreamWaitEvent(st, h->win_ev[(k - 2) % R][4], 0)); }      // window k - 2's dibit rows have left this device row block
        if (k == 0) { WINCHK(hipMemcpyAsync(d_anc, h_anc, C * sizeof(p25fe_anchor_t), hipMemcpyHostToDevice, st)); }
        WINCHK(hipEventRecord(ev[2], st));
        const size_t nb = p25fe_n_baseband_h(h, abs0, wn);
        const long view0 = (long)(h->abs_bb + nb_done) - h->look;
        p25fe_result_t* d_res = h->win_res.as<p25fe_result_t>() + (k % R) * C;
        uint8_t* d_dib = h->win_dib[b].as<uint8_t>();
        const PlanarGeo g(nb ? nb : 1);
        if (nb) {
            rc = launch_frontend(h, dev + SHARD_HALO * eb, fmt, stride, n_hist, wn, abs0, -(long)PLPAD - h->look, nullptr, 0, nullptr, st, &g);
            if (!rc) rc = launch_detect(h, nb, view0, st, rcall, false);
        }
        if (!rc) rc = launch_scan_slice(h, nb, view0, d_anc + (size_t)b * C, d_dib, dstride, nullptr, nullptr, 0, d_res, nb != 0, st, rcall, nb != 0);
        if (rc) { status = rc; break; }
        hipLaunchKernelGGL(k_anchors_from_results, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, st, d_res, d_anc + (size_t)(b ^ 1) * C, (int)C);
        if (k + 1 == n_win && nb) {                                  // the baseband tail for a later p25fe_slice on this handle
            ChunkRecvArgs cr;
            memset(&cr, 0, sizeof cr);
            cr.pl = planar_view(h, g); cr.n = (long)nb; cr.look = (int)h->look; cr.tail = d_tail;
            hipLaunchKernelGGL(k_tail_extract, dim3((unsigned)C), dim3(WV), 0, st, cr);
        }
        WINCHK(hipGetLastError());
        WINCHK(hipEventRecord(ev[3], st));
        // ---- results leave on their own stream
        WINCHK(hipStreamWaitEvent(os, ev[3], 0));
        WINCHK(hipMemcpyAsync(h_res[k % R], d_res, C * sizeof(p25fe_result_t), hipMemcpyDeviceToHost, os));
        WINCHK(hipMemcpyAsync(h_dib[b], d_dib, C * dstride, hipMemcpyDeviceToHost, os));
        WINCHK(hipEventRecord(ev[4], os));
        nb_done += nb;
        if (k >= 1) status = drain(k - 1);
    }
#undef WINCHK
    if (status == P25FE_OK) status = drain(n_win - 1);
    if (status != P25FE_OK) {
        // Nothing of the handle's STREAM STATE has moved (history, anchors, counters: the call can be repeated), and no copy or
        // kernel of this call is still in flight when it returns: they read the caller's capture and the handle's windows.
        // The caller's dibit buffer may already hold the dibits of earlier windows and n_dibits reads 0: treat both as undefined.
        (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(os);
        return status;
    }
    HIPCHK(h, hipStreamSynchronize(st));
    h->rs_n = 0;                                                     // a pending lock-drop list is consumed by a call that succeeded
    // ---- the handle's stream state after the capture
    const p25fe_result_t* last = h_res[(n_win - 1) % R];
    for (size_t c = 0; c < C; ++c) {
        n_dibits[c] = (size_t)total[c];
        h->anchor[c] = last[c].anchor_out;
        h->total_dibits[c] += total[c];
        // the last SHARD_HALO samples of [old history | capture]
        char* hist = h->hist_iq.data() + c * SHARD_HALO * 8;
        const char* srcc = static_cast<const char*>(iq) + c * n * eb;
        if (n >= SHARD_HALO) memcpy(hist, srcc + (n - SHARD_HALO) * eb, SHARD_HALO * eb);
        else { memmove(hist, hist + n * eb, (SHARD_HALO - n) * eb); memcpy(hist + (SHARD_HALO - n) * eb, srcc, n * eb); }
        const size_t nb_last = p25fe_n_baseband_h(h, h->abs_iq + (n_win - 1) * window, n - (n_win - 1) * window);   // (> 0: a window is at least 8 192 samples)
        float* t = h->tail_bb.data() + c * BBPAD;
        if (nb_last) {
            float merged[BBPAD];
            for (size_t j = 0; j < (size_t)BBPAD; ++j) {
                const long m = (long)nb_last - (long)BBPAD + (long)j;
                merged[j] = m >= -(long)(HIST_BB + h->look) ? h_tail[c * TAILN + j] : (j + nb_last < (size_t)BBPAD ? t[j + nb_last] : 0.0f);
            }
            memcpy(t, merged, sizeof merged);
        }
    }
    h->fmt_locked = fmt;
    h->abs_iq += n;
    h->abs_bb += nb_done;
    if (stats) {
        stats->n_windows = n_win;
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        stats->ms_h2d = ms_h2d; stats->ms_compute = ms_comp; stats->pinned_input = pinned ? 1 : 0;
    }
    return P25FE_OK;
}

int p25fe_nid_dev(p25fe_t* h, const uint8_t* d_dibits, size_t n_dibits, const uint64_t* d_sync_dibit,
                  const int64_t* d_sync_pos, size_t n_sync, p25fe_nid_t* d_out, void* stream)
{
    if (!h) return P25FE_ERR_ARG;
    if (n_sync == 0) return P25FE_OK;
    if (!d_dibits || !d_sync_dibit || !d_out) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_nid, dim3((unsigned)n_sync), dim3(256), 0, (hipStream_t)stream, d_dibits,
                       (unsigned long long)n_dibits, reinterpret_cast<const unsigned long long*>(d_sync_dibit),
                       reinterpret_cast<const long*>(d_sync_pos), d_out, (const p25fe_result_t*)nullptr, 0ull, 0ull);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

// host-buffer form for the file-driven harness: one channel's whole dibit stream and its sync events
int p25fe_nid(p25fe_t* h, const uint8_t* dibits, size_t n_dibits, const uint64_t* sync_dibit, const int64_t* sync_pos,
              size_t n_sync, p25fe_nid_t* out)
{
    if (!h) return P25FE_ERR_ARG;
    if (n_sync == 0) return P25FE_OK;                               // a capture without a frame sync: nothing to decode (empty vectors hand over null pointers)
    if ((!dibits && n_dibits) || !sync_dibit || !out) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    DevBuf d_dib, d_sd, d_sp, d_out;
    int rc = P25FE_OK;
    hipError_t e = d_dib.ensure(n_dibits + 16);
    if (e == hipSuccess) e = d_sd.ensure(n_sync * sizeof(uint64_t));
    if (e == hipSuccess) e = d_out.ensure(n_sync * sizeof(p25fe_nid_t));
    if (e == hipSuccess && sync_pos) e = d_sp.ensure(n_sync * sizeof(int64_t));
    if (e == hipSuccess && n_dibits) e = hipMemcpyAsync(d_dib.p, dibits, n_dibits, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sd.p, sync_dibit, n_sync * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && sync_pos) e = hipMemcpyAsync(d_sp.p, sync_pos, n_sync * sizeof(int64_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        rc = p25fe_nid_dev(h, d_dib.as<uint8_t>(), n_dibits, d_sd.as<uint64_t>(), sync_pos ? d_sp.as<int64_t>() : nullptr,
                           n_sync, d_out.as<p25fe_nid_t>(), h->stream);
        if (rc == P25FE_OK) e = hipMemcpyAsync(out, d_out.p, n_sync * sizeof(p25fe_nid_t), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    d_dib.release(); d_sd.release(); d_sp.release(); d_out.release();
    if (e != hipSuccess) { h->last_hip = (int)e; return P25FE_ERR_HIP; }
    return rc;
}

int p25fe_nid_batch_dev(p25fe_t* h, const uint8_t* d_dibits, size_t dibit_stride, const p25fe_result_t* d_result,
                        const uint64_t* d_sync_dibit, const int64_t* d_sync_pos, size_t sync_stride,
                        p25fe_nid_t* d_out, void* stream)
{
    if (!h || !d_dibits || !d_result || !d_sync_dibit || !d_out) return P25FE_ERR_ARG;
    if (sync_stride == 0) return P25FE_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_nid, dim3((unsigned)sync_stride, (unsigned)h->C), dim3(256), 0, (hipStream_t)stream, d_dibits,
                       0ull, reinterpret_cast<const unsigned long long*>(d_sync_dibit),
                       reinterpret_cast<const long*>(d_sync_pos), d_out, d_result, (unsigned long long)dibit_stride,
                       (unsigned long long)sync_stride);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_chan_stats_dev(p25fe_t* h, const p25fe_result_t* d_result, const p25fe_nid_t* d_nid, size_t sync_stride,
                         const float* d_power_dbm, p25fe_chan_stats_t* d_stats, void* stream)
{
    if (!h || !d_result || !d_stats) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_chan_stats, dim3((unsigned)h->C), dim3(256), 0, (hipStream_t)stream, d_result, d_nid,
                       (unsigned long long)sync_stride, d_power_dbm, d_stats);
    HIPCHK(h, hipGetLastError());
    return P25FE_OK;
}

int p25fe_profile_enable(p25fe_t* h, int on)
{
    if (!h) return P25FE_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (on && h->prof_ev.empty()) {
        try {
            h->prof_ev.assign((size_t)PROF_RING * 5, nullptr);
            h->prof_mask.assign((size_t)PROF_RING, 0);
        } catch (...) {
            h->prof_ev.clear();
            return P25FE_ERR_NOMEM;
        }
        for (auto& e : h->prof_ev) {
            const hipError_t er = hipEventCreate(&e);
            if (er != hipSuccess) {                                 // all or nothing: a half-made ring is never indexed
                for (auto& d : h->prof_ev) if (d) (void)hipEventDestroy(d);
                h->prof_ev.clear();
                h->last_hip = (int)er;
                return P25FE_ERR_HIP;
            }
        }
    }
    h->prof_on = on != 0;
    h->prof_level = on == 2 ? 2 : (on == 3 ? 3 : 1);
    h->prof_calls = 0;
    h->prof_seq = 0;
    return P25FE_OK;
}

int p25fe_profile_read(p25fe_t* h, double ms[4], uint64_t* n_calls)
{
    if (!h || !ms) return P25FE_ERR_ARG;
    for (int k = 0; k < 4; ++k) ms[k] = 0.0;
    const uint64_t calls = h->prof_calls;
    const uint64_t kept = calls < (uint64_t)PROF_RING ? calls : (uint64_t)PROF_RING;
    uint64_t n_k1 = 0;
    for (uint64_t s = 0; s < kept; ++s) {
        const unsigned m = h->prof_mask[s];                          // a slot holds the events its call recorded: all five
        if ((m & 3u) == 3u) ++n_k1;
        for (int k = 4; k >= 0; --k)                                 // (run_dev, shard pass 1) or 2..4 (shard pass 2)
            if (m & (1u << k)) { HIPCHK(h, hipEventSynchronize(h->prof_ev[s * 5 + k])); break; }
        for (int k = 0; k < 4; ++k) {
            if ((m & (3u << k)) != (3u << k)) continue;
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->prof_ev[s * 5 + k], h->prof_ev[s * 5 + k + 1]) != hipSuccess) {
                (void)hipGetLastError();                             // a slot whose launch had nothing to do (empty part of a shard)
                if (k == 0 && n_k1) --n_k1;
                continue;
            }
            ms[k] += t;
        }
    }
    if (n_calls) *n_calls = n_k1;          // calls that ran K1 (a shard's pass 2 has a slot of its own without one)
    h->prof_calls = 0;
    return P25FE_OK;
}

int p25fe_resync(p25fe_t* h)
{
    if (!h) return P25FE_ERR_ARG;
    for (auto& a : h->anchor) a.valid = 0;      // lock and, with it, the clock estimate (the next sync word starts from 10 / 1)
    return P25FE_OK;
}

int p25fe_resync_at_dev(p25fe_t* h, const int64_t* d_idx, size_t n_idx, size_t idx_stride)
{
    if (!h || (n_idx && !d_idx) || n_idx > 0x7fffffffu || (h->C > 1 && n_idx && idx_stride < n_idx)) return P25FE_ERR_ARG;
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    h->rs_idx = reinterpret_cast<const long*>(d_idx);
    h->rs_n = n_idx;
    h->rs_stride = idx_stride;
    return P25FE_OK;
}

// --------------------------------------------------------------------------------------------
// state blob: header | IQ history raw | baseband tail | anchors | totals
// --------------------------------------------------------------------------------------------
struct StateHeader {
    uint32_t magic, abi;
    int32_t n_channels, fmt_locked;
    uint64_t abs_iq, abs_bb;
};

int p25fe_state_size(const p25fe_t* h, size_t* n)
{
    if (!h || !n) return P25FE_ERR_ARG;
    const size_t C = (size_t)h->C;
    *n = sizeof(StateHeader) + C * SHARD_HALO * 8 + C * BBPAD * sizeof(float) + C * sizeof(p25fe_anchor_t) + C * sizeof(uint64_t);
    return P25FE_OK;
}

int p25fe_state_export(const p25fe_t* hc, void* buf, size_t cap, size_t* n)
{
    p25fe_t* h = const_cast<p25fe_t*>(hc);
    size_t need = 0;
    if (!h || !buf || p25fe_state_size(h, &need)) return P25FE_ERR_ARG;
    if (n) *n = need;
    if (cap < need) return P25FE_ERR_CAPACITY;
    const size_t C = (size_t)h->C;
    char* p = static_cast<char*>(buf);
    StateHeader hd = {STATE_MAGIC, P25FE_ABI_VERSION, h->C, h->fmt_locked, h->abs_iq, h->abs_bb};
    memcpy(p, &hd, sizeof hd); p += sizeof hd;
    memcpy(p, h->hist_iq.data(), C * SHARD_HALO * 8); p += C * SHARD_HALO * 8;       // the streaming state is host-side
    memcpy(p, h->tail_bb.data(), C * BBPAD * sizeof(float)); p += C * BBPAD * sizeof(float);
    memcpy(p, h->anchor.data(), C * sizeof(p25fe_anchor_t)); p += C * sizeof(p25fe_anchor_t);
    memcpy(p, h->total_dibits.data(), C * sizeof(uint64_t));
    return P25FE_OK;
}

int p25fe_state_import(p25fe_t* h, const void* buf, size_t n)
{
    size_t need = 0;
    if (!h || !buf || p25fe_state_size(h, &need) || n < need) return P25FE_ERR_ARG;
    const char* p = static_cast<const char*>(buf);
    StateHeader hd;
    memcpy(&hd, p, sizeof hd); p += sizeof hd;
    if (hd.magic != STATE_MAGIC || hd.abi != P25FE_ABI_VERSION || hd.n_channels != h->C) return P25FE_ERR_ARG;
    if (hd.fmt_locked != -1 && hd.fmt_locked != P25FE_FMT_U8 && hd.fmt_locked != P25FE_FMT_CF32) return P25FE_ERR_ARG;
    const size_t C = (size_t)h->C;
    {   // a blob is data from outside: every anchor's clock must be a usable period before anything is taken over
        const char* pa = p + C * SHARD_HALO * 8 + C * BBPAD * sizeof(float);
        for (size_t c = 0; c < C; ++c) {
            p25fe_anchor_t a;
            memcpy(&a, pa + c * sizeof a, sizeof a);
            // (0 / 0 reads as 10 / 1: include/p25fe.h)
            if (!(a.period_d == 0 && a.period_n == 0) && !clock_plausible(a.period_d, a.period_n)) return P25FE_ERR_ARG;
            if (!h->track && a.valid != 0 && !(a.period_d == 0 && a.period_n == 0) && !(a.period_d == SPS && a.period_n == 1)) return P25FE_ERR_ARG;
        }
    }
    memcpy(h->hist_iq.data(), p, C * SHARD_HALO * 8); p += C * SHARD_HALO * 8;
    memcpy(h->tail_bb.data(), p, C * BBPAD * sizeof(float)); p += C * BBPAD * sizeof(float);
    memcpy(h->anchor.data(), p, C * sizeof(p25fe_anchor_t)); p += C * sizeof(p25fe_anchor_t);
    memcpy(h->total_dibits.data(), p, C * sizeof(uint64_t));
    h->fmt_locked = hd.fmt_locked; h->abs_iq = hd.abs_iq; h->abs_bb = hd.abs_bb;
    h->rs_idx = nullptr; h->rs_n = 0; h->rs_stride = 0;             // (absolute indices of another stream position)
    return P25FE_OK;
}

P25FE_M_API_EXTRAS
}  // extern "C"
