// p25fe_recv.hip -- K2..K4: the symbol receiver (front half of MessageReceiver::feed, src/recv.rs:148-150, 204-210)
// on the polyphase baseband layout that K1 writes (PLPAD in p25fe_kernels.hip; included from there).
//
//   K2 k_detect : frame-sync detection (SPEC 3.7).  The 48-flop correlation is NOT evaluated at every sample: a sync
//                 candidate needs a normalised correlation >= sqrt(0.85) = 0.922 with the 24-symbol sign pattern,
//                 which forces at most 4 of the 24 symbol-spaced samples to have the wrong sign (proof below).  K1
//                 leaves the sign of every baseband sample in ten bit planes (1/32 of the baseband), so the screen is
//                 5 integer ops per sample on 3.6 MB instead of 48 flops per sample on 115 MB; the few positions that
//                 pass (0.08 % of random data) get the exact SPEC 3.7 arithmetic from the float planes, where a
//                 24-symbol window is 96 contiguous bytes.  Same detections as evaluating c / e everywhere, bit for bit.
//   K3          : the receiver's serial state (anchor in force, dibits / events so far) as a hierarchical scan over tile summaries by
//                 one-wave workgroups (k_scan_tiles; the general receiver: k_scan_tiles_g + k_scan_g_groups) -- see ScanArgs below.
//   K4 k_slice  : 4-level slicer (SPEC 3.8): a locked receiver reads ONE plane, contiguously.
//   k_planarize : linear baseband -> planes + sign bits, for the entry points that are handed a 48 kHz float stream
//                 (p25fe_slice / p25fe_slice_dev: the RecvEvent::Baseband hand-off of src/demod.rs:116).
//
// Why <= 4 sign mismatches.  Let v_j be the 24 window samples, g_j = +-1 the sync signs, c = sum g_j v_j,
// e = sum v_j^2.  Write v = a g + w with w orthogonal to g: c = 24 a, e = 24 a^2 + |w|^2.  A candidate has c > 0 and
// c^2 >= 20.4 e, i.e. |w|^2 <= 24 a^2 (24 / 20.4 - 1) = 4.24 a^2.  A sample whose sign disagrees with g_j (g_j v_j <= 0)
// has g_j w_j <= -a, so w_j^2 >= a^2: at most 4 of them fit.  The fp32 evaluation of c and e moves the ratio by
// < 1e-5 relative, nowhere near the 18 % that separates 4.24 from 5.  The screen counts a mismatch only where the
// SIGN BIT disagrees (a +0 under a '+' symbol is not counted), which can only under-count: still a superset.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace p25k {

constexpr int W = P25FE_PEAK_W;
constexpr int SPS = P25FE_SPS;
constexpr int NSYN = P25FE_SYNC_DIBITS;
constexpr int SYNC_SPAN = P25FE_SYNC_SPAN;                   // 230
constexpr int HIST_BB = SYNC_SPAN + 2 * W;                   // 240: baseband history the receiver needs
static_assert(SPS == SPS_ && PLPAD >= HIST_BB + P25FE_CLK_LOOKAHEAD, "planar layout matches the receiver");

constexpr int TSYM = 768;                                    // symbols of every plane per tile (24 words of sign bits)
constexpr int TS = TSYM * SPS;                               // 7680 baseband samples (decision indices) per tile
constexpr int TWORDS = TSYM / 32;                            // 24
constexpr int EVCAP = TS / (W + 1) + 8;                      // 1288: detections are at least W + 1 samples apart
constexpr unsigned SYNC_NEG_MASK = ~P25FE_SYNC_SIGN_MASK & 0xffffffu;   // bit j: sync symbol j (oldest first) is -3
constexpr int SCREEN_MAX_MISMATCH = 4;

struct Planar {             // blocked polyphase baseband of one call (PLPAD / planar_index in p25fe_kernels.hip)
    const float* f;         // channel 0: f[planar_index(i, r)] = b[10 i + r - PLPAD]
    long f_ch;              // floats per channel
    const uint32_t* bits;   // channel 0: bits[(i / 32) * 10 + r] bit i % 32 = sign of that sample
    long bits_ch;           // words per channel
};

struct TileRec {            // per (channel, tile) summary written by K2
    long first_event;       // absolute decision index e = s + W of the tile's first event, -1 if none
    long last_s;            // s of the tile's last event (absolute), valid if first_event >= 0
    float hi, mid, lo;      // thresholds of the last event
    int n_events;
    long post_count;        // instants in (first_event, tile_end) under the tile's own events
    int last_f;             // quarter-sample fraction of the last event's sync position (SPEC 3.8b; 0 outside the general receiver)
    int pad_;
};

// Packed per-tile summary for the scan (one coalesced 8-byte word per tile):
//   bits  0..12  first_off + 1   (0: the tile has no event)      bits 13..25  last_off + 1
//   bits 26..38  n_events                                          bits 39..51  post_count
constexpr int TS_BITS = 13;
static_assert(TS + 1 <= (1 << TS_BITS), "tile offsets fit the packed summary");
constexpr unsigned long long TS_MASK = (1ull << TS_BITS) - 1;
__host__ __device__ inline unsigned long long pack_tsum(int first_off, int last_off, int n_events, int post_count)
{
    return (unsigned long long)(first_off + 1) | ((unsigned long long)(last_off + 1) << TS_BITS) |
           ((unsigned long long)n_events << (2 * TS_BITS)) | ((unsigned long long)post_count << (3 * TS_BITS));
}

struct ScanOut {            // per (channel, tile) GROUP-LOCAL carry-in written by K3's group scan (completed by the slicer: group_fix)
    int src;                        // tile whose last event is in force at this tile's first sample; -1: the group's carry-in
    unsigned event_off;             // events of the GROUP before this tile
    unsigned long long dibit_off;   // dibits of the group before this tile (from its first own detection on)
};

// number of n in [lo, hi) with n > s and (n - s) % SPS == 0   (closed form)
__host__ __device__ inline long count_instants(long s, long lo, long hi)
{
    if (lo <= s) lo = s + 1;
    if (hi <= lo) return 0;
    const long k0 = (lo - s + SPS - 1) / SPS;     // first k with s + SPS*k >= lo
    const long k1 = (hi - 1 - s) / SPS;           // last k with s + SPS*k <= hi-1
    return k1 >= k0 ? k1 - k0 + 1 : 0;
}

// ------------------------------------------------------------------------------------------
// General receiver (SPEC 3.8b tracking symbol clock, and lock drops inside a range -- MessageReceiver::resync,
// src/recv.rs:136, 179, at given sample indices).  The fixed-stride receiver without lock drops keeps the kernels
// below as they were (packed 8-byte tile summaries, phase arithmetic modulo 10); everything else runs the *_g forms.
//
// A detection at s has a symbol clock: instants at s + floor(j D / N), j = 1, 2, ..., D / N = 10 / 1 or the interval from
// the previous sync word over its (rounded) symbol count.  It governs the instants with index in [s + W + 1, end),
// end = the next detection's decision index + 1 or the next lock drop.  In mode 1 the receiver runs L = 2 samples
// behind the baseband (the 4-tap interpolation around an instant in [i, i + 1) reads i - 1 .. i + 2): a range that
// owns baseband [a, b) processes indices [a - L, b - L) -- the kernels simply see a range that starts L samples
// earlier (K1 / k_planarize write sample m at planar position m + L + PLPAD); a lock drop before sample q kills the
// instants with index >= q - L.
// ------------------------------------------------------------------------------------------
constexpr int CLK_L = P25FE_CLK_LOOKAHEAD;

// Fraction of a sync position in quarter samples (SPEC 3.8b): vertex of the parabola through the correlation peak c0 = c[s] and
// its neighbours cl = c[s - 1], cr = c[s + 1]; -2 .. 2.  Only the PERIOD uses it; the anchor stays on the whole sample.
__host__ __device__ inline int sync_frac(float cl, float c0, float cr)
{
    const float num = cl - cr, den = (cl - (c0 + c0)) + cr;
    int f = 0;
    if (den < 0.0f) {
        const float q = (num * 0.5f) / den;
        if (q == q) {
            const float v = __builtin_floorf(q * 4.0f + 0.5f);
            f = v < -2.0f ? -2 : v > 2.0f ? 2 : (int)v;
        }
    }
    return f;
}
// the fraction travels in three bits (two's complement) of records that have them to spare
__host__ __device__ inline int frac3(unsigned v) { return (int)((v & 7u) ^ 4u) - 4; }

// usable: the interval passed the test, i.e. (D, N) IS a period estimate (also when it came out exactly nominal) -- SPEC 3.8c
__host__ __device__ inline void clock_period(bool track, bool prev_valid, long s_prev, int f_prev, long s_new, int f_new, int& D, int& N,
                                             bool* usable = nullptr)
{
    D = SPS; N = 1;
    if (usable) *usable = false;
    if (track && prev_valid) {
        const long dd = s_new - s_prev, nn = (dd + SPS / 2) / SPS;
        const long err = dd > SPS * nn ? dd - SPS * nn : SPS * nn - dd;
        if (nn >= 1 && dd <= (1L << P25FE_CLK_DMAX_LOG2) && (err << P25FE_CLK_TOL_SHIFT) <= SPS * nn) {
            if (usable) *usable = true;
            const long d4 = 4 * dd + (f_new - f_prev);
            // (an interval of exactly 10 N samples IS the nominal clock: kept as 10 / 1, the same instants, so that the
            // division-free paths below apply)
            if (d4 != 4 * SPS * nn) { D = (int)d4; N = (int)(4 * nn); }
        }
    }
}
// Could clock_period have produced (D, N)?  Anchors also arrive from outside (p25fe_state_import, a caller's d_anchor_in): a clock
// such as 1 / 2^30 would put 2^43 instants into a tile and wrap the 32-bit counts of the slicer.  Anything implausible is
// refused (import) or read as the nominal 10 / 1 (kernels).
__host__ __device__ inline bool clock_plausible(int D, int N)
{
    if (N == 1 && D == SPS) return true;
    if (N < 4 || (N & 3) != 0 || D <= 0) return false;
    const long nn = N / 4, d4 = D, nom = 4L * SPS * nn;
    if (d4 > 4 * (1L << P25FE_CLK_DMAX_LOG2) + 4) return false;
    const long err = d4 > nom ? d4 - nom : nom - d4;                 // quarter samples: 4 |dd - 10 nn| + |f_new - f_prev| <= 4 (10 nn >> shift) + 4
    return err <= 4 * ((SPS * nn) >> P25FE_CLK_TOL_SHIFT) + 4;
}
// number of instants j >= 1 of a clock (D, N) with floor(j D / N) < x
__host__ __device__ inline long clock_J(long x, int D, int N)
{
    if (x <= 0) return 0;
    if (N == 1 && D == SPS) return (x - 1) / SPS;                  // nominal clock: a division by a constant
    return (x * (long)N - 1) / (long)D;
}
// instants of the detection (s, D, N) with index in [lo, hi) that it governs (index > s + W)
__host__ __device__ inline long clock_count(long s, int D, int N, long lo, long hi)
{
    if (lo < s + W + 1) lo = s + W + 1;
    if (hi <= lo) return 0;
    return clock_J(hi - s, D, N) - clock_J(lo - s, D, N);
}

struct RecvOpt {
    int track;                  // SPEC 3.8b
    int n_resync;               // lock drops per channel
    const long* resync;         // [ch][resync_stride] ascending absolute baseband indices q (lock dropped before sample q); nullable
    long resync_stride;
};
// first kill index f = q - L >= x of a channel's lock drops (LONG_MAX if none); L = the mode's lookahead
__device__ __forceinline__ long first_kill(const RecvOpt& o, int ch, long x)
{
    if (o.n_resync == 0) return 0x7fffffffffffffffL;
    const long* r = o.resync + (size_t)ch * o.resync_stride;
    const long key = x + (o.track ? CLK_L : 0);
    int lo = 0, hi = o.n_resync;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (r[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < o.n_resync ? r[lo] - (o.track ? CLK_L : 0) : 0x7fffffffffffffffL;
}

// Per-tile summary of the general receiver (K2 -> K3 / K4), 32 bytes
struct alignas(8) TileSumG {
    unsigned pre_end1;          // (tile offset from which the carry-in anchor no longer governs) + 1; 0: the tile has no event of any kind
    unsigned first1;            // first own detection's decision offset + 1 (bits 0..15; 0: none) | its sync position's fraction (3 bits) << 16
    unsigned end0;              // tile offset (exclusive) at which the first detection's governed interval ends
    unsigned last1;             // last own detection's decision offset + 1
    unsigned n_det_flags;       // detections (low 16 bits) | flags << 16
    unsigned post_rest;         // instants governed by own detections 1 .. n_det - 1 (their clocks are known inside the tile)
    int out_D, out_N;           // clock of the last detection, if G_OUT_PERIOD_KNOWN
};
constexpr unsigned G_FIRST_TRACKS = 1u;       // no lock drop between the tile's start and its first detection
constexpr unsigned G_OUT_VALID = 2u;          // the tile ends locked on its last detection
constexpr unsigned G_OUT_PERIOD_KNOWN = 4u;   // ... whose clock does not depend on the carry-in

__device__ __forceinline__ unsigned char slice_dibit(float v, float hi, float mid, float lo)
{
    return v >= hi ? 1 : v >= mid ? 0 : v >= lo ? 2 : 3;
}

// The 24 symbol-spaced samples v_j, j = 0 (oldest) .. 23, of the sync word whose last symbol is planar sample p
// (p = m + PLPAD): consecutive symbols of plane p % 10 -- contiguous inside a block, 288 floats further in the next.
__device__ __forceinline__ void sync_gather(const float* __restrict__ f, long p, float (&v)[NSYN])
{
    const long i = p / SPS;
    const int r = (int)(p - i * SPS);
    const long first = i - (NSYN - 1);
    const float* base = f + planar_index(first, r);
    const int n0 = 32 - (int)(first & 31);                      // samples left in the first block
#pragma unroll
    for (int j = 0; j < NSYN; ++j) v[j] = base[j + (j >= n0 ? PL_BLK - 32 : 0)];
}

// SPEC 3.7 on one window
__device__ __forceinline__ void sync_corr(const float (&w)[NSYN], float& c, float& e)
{
    float cc = 0.f, ee = 0.f;
#pragma unroll
    for (int j = 0; j < NSYN; ++j) {
        const float x = w[j];
        cc = ((P25FE_SYNC_SIGN_MASK >> j) & 1u) ? cc + x : cc - x;
        ee = __builtin_fmaf(x, x, ee);
    }
    c = cc; e = ee;
}

// SPEC 3.8: thresholds from the sync word's own levels
__device__ __forceinline__ void sync_thresholds(const float (&w)[NSYN], float& hi, float& mid, float& lo)
{
    float Pp = 0.f, Nn = 0.f;
#pragma unroll
    for (int j = 0; j < NSYN; ++j) {
        const float v = w[j];
        if ((P25FE_SYNC_SIGN_MASK >> j) & 1u) Pp = Pp + v; else Nn = Nn + v;
    }
    Pp = Pp * P25FE_SYNC_INV_NPOS;
    Nn = Nn * P25FE_SYNC_INV_NNEG;
    mid = (Pp + Nn) * 0.5f;
    const float span = (Pp - Nn) * 0.5f;
    const float d = span * P25FE_SLICE_FRAC;
    hi = mid + d;
    lo = mid - d;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int wave_incl_sum(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_incl_sum64(unsigned long long v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ int lane_rank(unsigned long long mask)      // set bits of mask below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// A record another workgroup of the SAME launch reads (K3's group aggregates): written through (sc1: 8-byte relaxed agent-scope stores), so
// that no release fence is needed -- a fence is `buffer_wbl2`, a write-back of the XCD's whole L2 (with one in front of every ticket of a
// first version, 3 750 per launch, K2 was four times slower and K1 beside it 30 % slower: measured, round 6).  The writer drains (`s_waitcnt vmcnt(0)`) before it takes its ticket (last_arrival);
// the reader's acquire drops its CU's stale L1 lines.  cdna_hip_programming.md guideline 16, recipe R1.
template <class T> __device__ __forceinline__ void publish(T* dst, const T& v)
{
    static_assert(sizeof(T) % 8 == 0 && alignof(T) >= 8, "whole 8-byte words");
    unsigned long long w[sizeof(T) / 8];
    __builtin_memcpy(w, &v, sizeof(T));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 8; ++k) __hip_atomic_store(d + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// K3, the receiver's serial state, as work of ONE-WAVE workgroups (round 6).  It was a scan by one workgroup of 512 threads behind K2
// (k_scan: 33 KB of LDS; the general receiver's k_scan_g: 48 - 53 KB): beside a running front end -- twelve one-wave workgroups on every
// CU, a freed slot refilled at once -- such a workgroup finds no CU with room until K1 DRAINS (p90 214 us in round 5's pipelined trace
// against 9.4 us alone), so the receive chain of a pipelined call ran at the tail of the next call's K1 instead of beside it, and the
// tracking clock's longer chain did not fit the K1 period (0.34 - 0.35 ms per step in mode 2).  Now a hierarchical scan:
//   * tiles form groups of GT = 64 (one tile per lane).  A group is scanned with NO carry-in by one wave: per tile the group-local
//     carry-in record (ScanOut), per group an aggregate (GroupAgg), published write-through;
//   * the last group of a channel to arrive (a ticket) scans the aggregates -- 64 groups per step, the carried state uniform -- under
//     the range's carry-in anchor: per group the anchor in force at its first sample and the dibits / detections in front of it
//     (GroupPre), and the range's result record;
//   * every slicer workgroup turns its tile's group-local record into the final one in closed form (group_fix: the rule ShardFix
//     applies one level up): tiles in front of the group's first own detection are governed by the group's carry-in, every later
//     tile's offset grows by the instants that carry-in governs inside the group.
// "Latest anchor wins" and event-free stretches counted in closed form -- the same algebra as k_scan, k_scan's results bit for bit.
// WHERE the group scans run: in launches of their own, one wave per group -- k_scan_tiles (fixed stride: 51 - 59 VGPRs, no LDS, 5.8 us
// alone, p90 17 us beside K1), k_scan_tiles_g + k_scan_g_groups (general receiver: GroupSumG below).  Running them in K2's own TAIL
// instead (the last detection workgroup of a group scans it) was built and measured for both receivers: the fixed stride lost by it --
// every one of 3 750 detection workgroups then lives ~4 us longer (drain of its write-through stores + the ticket's round trip) and K1
// beside them paid 10 - 25 us (profiles/r06_rx_grid_ab.txt) --, the general receiver neither won nor lost (profiles/r06_tree_ab.txt):
// K2 stays pure detection.  k_range_scan / k_range_scan_g are the top step alone: the re-scan of a time shard under a resolved carry-in
// (p25fe_shard_pass2), and the record of an empty range.
// ------------------------------------------------------------------------------------------
constexpr int GT = 64;                                           // tiles per group (one per lane)
#ifndef P25FE_HEAD_WAIT_TICKS
#define P25FE_HEAD_WAIT_TICKS 200000000ull                       // 2 s of the 100 MHz wall clock (tests build a short one)
#endif
constexpr unsigned long long HEAD_WAIT_TICKS = P25FE_HEAD_WAIT_TICKS;
struct GroupAgg {               // per (channel, group), its tiles alone
    long first_event;           // absolute decision index of the group's first detection, -1: none
    long last_s;                // position of its last detection
    unsigned long long after_first;     // instants in (first_event, group end) under the group's own detections
    int src;                    // tile of the last detection
    unsigned n_events;
};
struct GroupPre {               // per (channel, group), from the range's point of view
    long s;                     // the anchor in force at the group's first sample ...
    unsigned long long dibit_off;       // dibits of the range in front of the group
    unsigned long long carry_cnt;       // instants that anchor governs inside the group ([group start, first_event] or all of it)
    int src;                    // ... tile whose record holds its thresholds; -1: the range's anchor_in
    int valid;                  // ... 0: not locked
    unsigned event_off;         // detections of the range in front of the group
    unsigned pad_;
};
struct ScanArgs {               // K3, fixed-stride receiver (k_scan_tiles / k_range_scan)
    const unsigned long long* tsum;     // [ch][n_tiles] K2's packed summaries
    const TileRec* recs;        // [ch][n_tiles]
    int n_tiles;
    long n;                     // owned samples per channel
    long abs0;
    ScanOut* outs;              // [ch][n_tiles] group-local carry-ins
    GroupAgg* gagg;             // [ch][n_groups]
    GroupPre* gpre;             // [ch][n_groups]
    unsigned* tickets;          // [ch][n_groups + 1], zero between launches (the last arrival resets its counter)
    const p25fe_anchor_t* anchor_in;    // nullable, [ch]
    p25fe_result_t* result;     // [ch]
    unsigned long long n_baseband;
};
__host__ __device__ inline int n_groups_of(int n_tiles) { return (n_tiles + GT - 1) / GT; }

// ------------------------------------------------------------------------------------------
// K2: frame-sync detection.  One wave per tile of TS decision indices e = s + W (tile t owns the detections DECIDED in
// [TS t, TS t + tn): every dependency points left).  Lane (plane r = lane / 6, block = lane % 6) screens 128 symbol
// positions of its plane from five 32-bit words of sign bits.
// ------------------------------------------------------------------------------------------
struct DetArgs {
    Planar pl;
    long n;                 // owned baseband samples per channel
    long abs0;              // absolute index of owned sample 0
    int n_tiles;
    TileRec* recs;          // [ch][n_tiles]
    unsigned long long* tsum;   // [ch][n_tiles]
    uint16_t* evl;          // [ch][n_tiles][EVCAP] decision offsets of the tile's detections, ascending
    float* evthr;           // [ch][n_tiles][EVTHR_N][3] slicer thresholds (hi, mid, lo) of the tile's first EVTHR_N detections
    // general receiver only (k_detect<true>)
    RecvOpt opt;
    TileSumG* gsum;         // [ch][n_tiles]
    uint32_t* evg;          // [ch][n_tiles][EVCAP] per detection: end offset of its governed interval | (tracks its predecessor) << 15 | fraction of its sync position (3 bits) << 16
    // time shards whose head segment runs on ANOTHER stream (p25fe_shard_pass1_head): the tiles that read the head's planes
    // (tile <= head_tile_max) wait for the word the head launch's last workgroup writes (K1Args.done_flag) -- no cross-stream
    // event wait in front of this launch (that wait costs ~10 - 20 us of an otherwise idle GPU)
    const unsigned* head_flag;  // nullable
    unsigned head_seq;
    int head_tile_max;
    unsigned* head_err;         // with head_flag: receives head_seq if a wait gave up (p25fe_shard_head_check)
};

constexpr int EVTHR_N = 4;                                       // detections per tile whose thresholds K2 hands to K4 (more: K4 recomputes)
constexpr int K2_DCAP = 64;                                      // detections per tile whose thresholds K2 remembers (more: it recomputes)
constexpr int K2_LANES = 60;                                     // 10 planes x 6 blocks of 4 words
constexpr int K2_HCAP = 2048;                                    // screened positions awaiting the exact test (256 -- 3.6 KB of LDS less
                                                                 // -- measured in round 6 on the fixed-stride kernel: no difference)

template <bool GEN> __device__ __forceinline__ void detect_tile(const DetArgs& a, const int tile, const int ch)
{
    // LDS: what the exact test needs (screened positions, the current round's candidates) is dead when the sorted list is made, so the two
    // phases share their bytes -- 7.1 KB per workgroup (general receiver: 8.4 KB) instead of 9.7 (16.1): beside a running front end a
    // workgroup of the chain gets the LDS one retiring K1 workgroup frees plus the CU's spare, and a K1 workgroup wants the same bytes back.
    // (The general detection with 21 KB waited for K1 to drain: 250 us instead of 20 alone; with 16 KB it still took 100 - 200, now 45.)
    struct Phase1 {
        uint16_t hits[K2_HCAP];
        uint16_t cands[WV];
        float cn[5][12];
        // A candidate's thresholds (SPEC 3.8) come from the same 24 samples as its correlation: computed where those are in
        // registers and remembered for the detections, instead of gathered again (a dependent round trip) for the summary.
        float cthr[WV][3];                                       // per candidate of the current round
    };
    struct Phase2 {
        uint16_t evs[EVCAP];                                     // the tile's detections, sorted list of decision offsets
        uint16_t endo[GEN ? EVCAP : 1];                          // GEN: where each detection's governed interval ends
    };
    union Arena { Phase1 p1; Phase2 p2; };
    __shared__ Arena AR;
    auto& HITS = AR.p1.hits;
    auto& CANDS = AR.p1.cands;
    auto& CN = AR.p1.cn;
    auto& CTHR = AR.p1.cthr;
    auto& EVS = AR.p2.evs;
    __shared__ unsigned EVB[TS / 32];                            // detections of the tile, bit = decision offset
    __shared__ uint16_t DETE[K2_DCAP];                           // per detection: decision offset ...
    __shared__ float DETT[K2_DCAP][3];                           // ... and thresholds
    __shared__ unsigned DETN;
    // GEN: the detection's position fraction (3 bits), one byte per BLOCK of W + 1 decision offsets: detections are more than W samples
    // apart, so a block never holds two, and a byte is written whole (no clearing pass, no read-modify-write); read only at detections.
    static_assert(TS % (W + 1) == 0, "fraction table: whole blocks");
    __shared__ uint8_t FRO[GEN ? TS / (W + 1) : 4];
    auto fro_set = [&](int off, int v) { FRO[off / (W + 1)] = (uint8_t)(v & 7); };
    auto fro_get = [&](int off) -> unsigned { return (unsigned)FRO[off / (W + 1)] & 7u; };
    const int lane = threadIdx.x;
    const float* f = a.pl.f + (size_t)ch * a.pl.f_ch;
    const long t0 = (long)tile * TS;
    const int tn = a.n - t0 < TS ? (int)(a.n - t0) : TS;
    const long pbase = t0 + PLPAD - W;                           // planar index of the position decided at tile offset 0

    for (int k = lane; k < TS / 32; k += WV) EVB[k] = 0u;
    if (lane == 0) DETN = 0u;

    // ---- screen: hit words, bit u of hw[j] = position (word step j, bit u) has <= 4 sign mismatches
    unsigned hw[4] = {0u, 0u, 0u, 0u};
    const int r = lane / 6, blk = lane % 6;
    const int dlt = r >= SPS / 2 ? 1 : 0;                        // planes 5..9 start one symbol earlier (PLPAD - W = 315)
    if (lane < K2_LANES) {
        const uint32_t* bw = a.pl.bits + (size_t)ch * a.pl.bits_ch + ((size_t)tile * TWORDS + 4 * blk) * SPS + r;
        unsigned wd[6];
#pragma unroll
        for (int k = 0; k < 5; ++k) wd[k] = bw[SPS * k];
        wd[5] = 0u;
        if (dlt) {                                                // shift the 160-bit string up by one: same code for both halves
#pragma unroll
            for (int k = 4; k >= 1; --k) wd[k] = __builtin_amdgcn_alignbit(wd[k], wd[k - 1], 31);
            wd[0] <<= 1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned acc = 0u;
#pragma unroll
            for (int u = 31; u >= 0; --u) {
                // position = local bit 32 (j + 1) + u; its window = bits [pos - 23, pos] = 24 bits from bit 9 + u of word j
                const unsigned win = u <= 22 ? __builtin_amdgcn_alignbit(wd[j + 1], wd[j], 9 + u)
                                             : __builtin_amdgcn_alignbit(wd[j + 2], wd[j + 1], u - 23);
                const unsigned x = (win ^ SYNC_NEG_MASK) & 0xffffffu;
                const int t = __builtin_popcount(x) - (SCREEN_MAX_MISMATCH + 1);     // negative <=> screened in
                acc = __builtin_amdgcn_alignbit(acc, (unsigned)t, 31);                // acc = acc << 1 | sign(t)
            }
            hw[j] = acc;
        }
    }

    P25FE_M_DET_CUT_SCREEN;                                         // (measurement builds: the sign-bit screen alone)
    // ---- exact test of the screened positions (SPEC 3.7), candidates -> peak test against their 10 neighbours
    int nh = 0;
    auto flush_hits = [&]() {
        for (int b = 0; b < nh; b += WV) {
            const int h = b + lane;
            const bool act = h < nh;
            const int eo = act ? (int)HITS[h] : 0;
            bool cand = false;
            float th_hi = 0.f, th_mid = 0.f, th_lo = 0.f;
            if (act) {
                float c, e, v[NSYN];
                sync_gather(f, pbase + eo, v);
                sync_corr(v, c, e);
                cand = (c > 0.0f) && (e >= P25FE_SYNC_E_MIN) && (c * c >= P25FE_SYNC_RHO2_N * e);
                if (cand) sync_thresholds(v, th_hi, th_mid, th_lo);
            }
            unsigned long long cm = __ballot(cand);
            if constexpr (P25FE_M_DET_NO_PEAK_TEST) { cm = 0ull; if (cand) EVB[eo >> 5] = 1u; }      // (measurement builds: no peak test)
            const int nc = __popcll(cm);
            if (cand) {
                const int slot = lane_rank(cm);
                CANDS[slot] = (uint16_t)eo;
                CTHR[slot][0] = th_hi; CTHR[slot][1] = th_mid; CTHR[slot][2] = th_lo;
            }
            phase_sync();
            for (int cb = 0; cb < nc; cb += 5) {
                const int q = lane / 11, d = lane - 11 * q;
                const bool act2 = lane < 55 && cb + q < nc;
                const int ec = act2 ? (int)CANDS[cb + q] : 0;
                if (act2) {
                    float c, e, v[NSYN];
                    sync_gather(f, pbase + ec + d - W, v);
                    sync_corr(v, c, e);
                    CN[q][d] = c;
                }
                phase_sync();
                if (act2 && d == W) {
                    const float c0 = CN[q][W];
                    bool det = true;
#pragma unroll
                    for (int i = 1; i <= W; ++i) det = det && (c0 > CN[q][W - i]) && (c0 >= CN[q][W + i]);
                    if (det) {
                        if constexpr (GEN) fro_set(ec, sync_frac(CN[q][W - 1], c0, CN[q][W + 1]));
                        atomicOr(&EVB[ec >> 5], 1u << (ec & 31));
                        const unsigned di = atomicAdd(&DETN, 1u);
                        if (di < (unsigned)K2_DCAP) {
                            DETE[di] = (uint16_t)ec;
                            DETT[di][0] = CTHR[cb + q][0]; DETT[di][1] = CTHR[cb + q][1]; DETT[di][2] = CTHR[cb + q][2];
                        }
                    }
                }
                phase_sync();
            }
        }
        nh = 0;
    };
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        unsigned hwj = hw[j];
        while (true) {
            const bool any = hwj != 0u;
            if (__ballot(any) == 0ull) break;
            int eo = TS;
            if (any) {
                const int u = __builtin_ctz(hwj);
                hwj &= hwj - 1u;
                eo = SPS * (128 * blk + 32 * (j + 1) + u - dlt) + r - (PLPAD - W);
            }
            const bool keep = eo < tn;                              // positions past the end of the range are not decided here
            const unsigned long long km = __ballot(keep);
            if (keep) HITS[nh + lane_rank(km)] = (uint16_t)eo;
            nh += __popcll(km);
            phase_sync();
            if (nh > K2_HCAP - WV) flush_hits();
        }
    }
    flush_hits();
    phase_sync();

    // ---- sorted event list + tile summary
    int cnt = 0;
    unsigned ew[4] = {0u, 0u, 0u, 0u};
    if (lane < TS / 128) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ew[k] = EVB[4 * lane + k]; cnt += __builtin_popcount(ew[k]); }
    }
    const int incl = wave_incl_sum(cnt, lane);
    const int n_ev = __shfl(incl, WV - 1, 64);
    long kill0 = 0x7fffffffffffffffL;                            // GEN: first lock drop at or after the tile's start
    if constexpr (GEN) kill0 = first_kill(a.opt, ch, a.abs0 + t0);
    if (n_ev == 0) {
        if (lane == 0) {
            TileRec rc;
            rc.first_event = -1; rc.last_s = -1; rc.hi = rc.mid = rc.lo = 0.f; rc.n_events = 0; rc.post_count = 0; rc.last_f = 0; rc.pad_ = 0;
            a.recs[(size_t)ch * a.n_tiles + tile] = rc;
            a.tsum[(size_t)ch * a.n_tiles + tile] = 0ull;
            if constexpr (GEN) {
                TileSumG g;
                g.pre_end1 = kill0 < a.abs0 + t0 + tn ? (unsigned)(kill0 - (a.abs0 + t0)) + 1u : 0u;   // a lock drop alone ends the carry-in
                g.first1 = 0u; g.end0 = 0u; g.last1 = 0u; g.n_det_flags = 0u; g.post_rest = 0u; g.out_D = SPS; g.out_N = 1;
                a.gsum[(size_t)ch * a.n_tiles + tile] = g;
            }
        }
        return;
    }
    {
        int pos = incl - cnt;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned w_ = ew[k];
            while (w_) {
                const int u = __builtin_ctz(w_);
                w_ &= w_ - 1u;
                EVS[pos++] = (uint16_t)(128 * lane + 32 * k + u);
            }
        }
    }
    phase_sync();
    uint16_t* evl = a.evl + ((size_t)ch * a.n_tiles + tile) * EVCAP;
    int post = 0;
    for (int k = lane; k < n_ev; k += WV) {
        const int ek = EVS[k];
        evl[k] = (uint16_t)ek;
        const int nxt = k + 1 < n_ev ? (int)EVS[k + 1] + 1 : tn;
        post += (int)count_instants(ek - W, ek + 1, nxt);          // instants in (e_k, e_{k+1}] under anchor s_k
    }
    post = wave_sum_i(post);
    const int first_off = EVS[0], last_off = EVS[n_ev - 1];
    if constexpr (GEN) {
        // per detection k: where its governed interval ends (the next detection's decision index + 1, the next lock drop or
        // the tile's end) and whether detection k + 1 may take its period from the interval s_k -> s_{k+1} (no drop between)
        auto& ENDO = AR.p2.endo;
        const long T0 = a.abs0 + t0, TE = T0 + tn;
        for (int k = lane; k < n_ev; k += WV) {
            const long ek = T0 + EVS[k];
            const long f = first_kill(a.opt, ch, ek + 1);
            const long nxt = k + 1 < n_ev ? T0 + EVS[k + 1] + 1 : TE;
            ENDO[k] = (uint16_t)((f < nxt ? f : nxt) - T0);
        }
        phase_sync();
        // detection k may take its period from the interval s_{k-1} -> s_k when no lock drop ended k - 1's governed interval before
        // k's decision, i.e. when that interval ends AT k's decision index + 1 (k = 0: no drop between the tile's start and it)
        auto trk = [&](int k) -> bool { return k == 0 ? (kill0 > T0 + first_off) : ((int)ENDO[k - 1] == (int)EVS[k] + 1); };
        uint32_t* evg = a.evg + ((size_t)ch * a.n_tiles + tile) * EVCAP;
        long rest = 0;
        for (int k = lane; k < n_ev; k += WV) {
            evg[k] = (uint32_t)ENDO[k] | ((uint32_t)(trk(k) ? 1u : 0u) << 15) | ((uint32_t)fro_get(EVS[k]) << 16);
            if (k >= 1) {
                const long sk = T0 + EVS[k] - W, sp = T0 + EVS[k - 1] - W;
                int D, N;
                clock_period(a.opt.track != 0, trk(k), sp, frac3(fro_get(EVS[k - 1])), sk, frac3(fro_get(EVS[k])), D, N);
                rest += clock_count(sk, D, N, sk + W + 1, T0 + ENDO[k]);
            }
        }
        rest = (long)wave_sum_i((int)rest);
        if (lane == 0) {
            TileSumG g;
            const long pe = kill0 < T0 + first_off + 1 ? kill0 : T0 + first_off + 1;
            g.pre_end1 = (unsigned)(pe - T0) + 1u;
            g.first1 = ((unsigned)first_off + 1u) | ((unsigned)fro_get(first_off) << 16);
            g.end0 = ENDO[0];
            g.last1 = (unsigned)last_off + 1u;
            unsigned fl = trk(0) ? G_FIRST_TRACKS : 0u;
            if ((int)ENDO[n_ev - 1] == tn) fl |= G_OUT_VALID;
            g.out_D = SPS; g.out_N = 1;
            if (n_ev >= 2) {
                fl |= G_OUT_PERIOD_KNOWN;
                clock_period(a.opt.track != 0, trk(n_ev - 1), T0 + EVS[n_ev - 2] - W, frac3(fro_get(EVS[n_ev - 2])), T0 + last_off - W,
                             frac3(fro_get(last_off)), g.out_D, g.out_N);
            } else if (!a.opt.track || !trk(0)) {
                fl |= G_OUT_PERIOD_KNOWN;                          // nominal period: nothing to take it from
            }
            g.n_det_flags = (unsigned)n_ev | (fl << 16);
            g.post_rest = (unsigned)rest;
            a.gsum[(size_t)ch * a.n_tiles + tile] = g;
        }
    }
    // thresholds of the first EVTHR_N detections (for K4) and of the last one (the anchor the tile hands on): looked up in
    // the remembered table, recomputed from the planes only if the tile had more detections than the table holds
    const int n_tab = DETN < (unsigned)K2_DCAP ? (int)DETN : K2_DCAP;
    auto thresholds_of = [&](int eoff, float& h, float& m, float& l) {
        const unsigned long long hit = __ballot(lane < n_tab && (int)DETE[lane < n_tab ? lane : 0] == eoff);
        if (hit) {                                                   // uniform
            const int i = __builtin_ctzll(hit);
            h = DETT[i][0]; m = DETT[i][1]; l = DETT[i][2];
        } else {
            float v[NSYN];
            sync_gather(f, pbase + eoff, v);                         // uniform: every lane, same window
            sync_thresholds(v, h, m, l);
        }
    };
    float hi, mid, lo;
    thresholds_of(last_off, hi, mid, lo);
    float* evthr = a.evthr + ((size_t)ch * a.n_tiles + tile) * (EVTHR_N * 3);
    for (int k = 0; k < EVTHR_N && k < n_ev; ++k) {
        float h, m, l;
        thresholds_of((int)EVS[k], h, m, l);
        if (lane == 0) { evthr[3 * k] = h; evthr[3 * k + 1] = m; evthr[3 * k + 2] = l; }
    }
    if (lane == 0) {
        TileRec rc;
        rc.first_event = a.abs0 + t0 + first_off;
        rc.last_s = a.abs0 + t0 + last_off - W;
        rc.hi = hi; rc.mid = mid; rc.lo = lo;
        rc.n_events = n_ev;
        rc.post_count = post;
        rc.last_f = 0; rc.pad_ = 0;
        if constexpr (GEN) rc.last_f = frac3(fro_get(last_off));
        a.recs[(size_t)ch * a.n_tiles + tile] = rc;
        a.tsum[(size_t)ch * a.n_tiles + tile] = pack_tsum(first_off, last_off, n_ev, post);
    }
}

// Is this workgroup the last of `expected` to arrive at `counter`?  What the arrivals PUBLISHED before (write-through stores, drained
// here in front of the ticket) is visible to the last one behind its acquire; the last arrival leaves the counter at zero for the next launch.
__device__ __forceinline__ bool last_arrival(unsigned* counter, unsigned expected)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have left
    unsigned ticket = 0u;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket != expected - 1u) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
__device__ __forceinline__ long shfl_l(long v, int src) { return (long)__shfl((unsigned long long)v, src, 64); }
// lane j's value, j uniform: v_readlane_b32 into scalar registers, no LDS crossbar
__device__ __forceinline__ int rdl(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ long rdl_l(long v, int j)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xffffffffull), j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), j);
    return (long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float rdl_f(float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); }
// The slicer's view of its tile's detections: one per lane, 64 at a time, read with v_readlane under a uniform index -- no LDS (K4 rides in
// whatever a running front end leaves free), and the loads leave together with the tile's records instead of one round trip later.
struct EvLanes {
    unsigned ev;                // evl[lane]: decision offset of detection `lane`
    unsigned eg;                // evg[lane] (general receiver)
    float th;                   // evthr[lane], lane < 3 EVTHR_N: thresholds of the tile's first detections
};
__device__ __forceinline__ EvLanes ev_lanes_load(const uint16_t* evl, const uint32_t* evg, const float* evthr)
{
    EvLanes e;
    e.ev = evl[threadIdx.x];                                         // (the lists are EVCAP >= 64 entries long whatever the tile holds)
    e.eg = evg ? evg[threadIdx.x] : 0u;
    e.th = threadIdx.x < EVTHR_N * 3 ? evthr[threadIdx.x] : 0.f;
    return e;
}
static_assert(EVCAP >= WV && EVTHR_N * 3 <= WV, "EvLanes: one batch of the list, the thresholds in one register");
__device__ __forceinline__ int last_set_below(unsigned long long mask, int lane)      // highest set bit of mask below `lane`, -1: none
{
    const unsigned long long m = mask & ((1ull << lane) - 1ull);
    return m ? 63 - __builtin_clzll(m) : -1;
}

// One group of tiles, no carry-in: lane = tile.
__device__ __forceinline__ void group_scan(const ScanArgs& a, const int g, const int ch)
{
    const ScanArgs& t = a;
    const int lane = threadIdx.x, tl = g * GT + lane;
    const bool act = tl < a.n_tiles;
    const unsigned long long u = act ? a.tsum[(size_t)ch * a.n_tiles + tl] : 0ull;
    const int first1 = (int)(u & TS_MASK);
    const bool has = first1 != 0;
    const unsigned long long evm = __ballot(has);
    const int pl = last_set_below(evm, lane);                       // latest tile of the group with a detection, in front of mine
    const unsigned long long up = __shfl(u, pl < 0 ? 0 : pl, 64);
    const long T0 = a.abs0 + (long)tl * TS;
    const long rem = a.n - (long)tl * TS;
    const int tn = rem < TS ? (int)rem : TS;
    unsigned pre = 0u;                                              // instants of my tile under the anchor it starts with, if that is the group's own
    if (act && pl >= 0) {
        const long sp = a.abs0 + (long)(g * GT + pl) * TS + ((long)((up >> TS_BITS) & TS_MASK) - 1) - W;
        pre = (unsigned)count_instants(sp, T0, T0 + (has ? first1 : tn));       // (the instant AT the decision index is still the old anchor's)
    }
    const unsigned long long cnt = act ? (unsigned long long)pre + ((u >> (3 * TS_BITS)) & TS_MASK) : 0ull;
    const unsigned long long ev = (u >> (2 * TS_BITS)) & TS_MASK;
    const unsigned long long mine = cnt | (ev << 40);               // (64 tiles: < 2^18 dibits, < 2^17 detections)
    const unsigned long long incl = wave_incl_sum64(mine, lane);
    if (act) {
        ScanOut o;
        o.src = pl >= 0 ? g * GT + pl : -1;
        o.event_off = (unsigned)((incl - mine) >> 40);
        o.dibit_off = (incl - mine) & ((1ull << 40) - 1);
        t.outs[(size_t)ch * a.n_tiles + tl] = o;
    }
    const unsigned long long tot = __shfl(incl, WV - 1, 64);
    const int fl = evm ? __builtin_ctzll(evm) : 0, ll = evm ? 63 - __builtin_clzll(evm) : 0;
    const unsigned long long uf = __shfl(u, fl, 64), ul = __shfl(u, ll, 64);
    if (lane == 0) {
        GroupAgg A;
        A.first_event = -1; A.last_s = 0; A.src = -1;
        if (evm) {
            A.first_event = a.abs0 + (long)(g * GT + fl) * TS + (long)(uf & TS_MASK) - 1;
            A.last_s = a.abs0 + (long)(g * GT + ll) * TS + ((long)((ul >> TS_BITS) & TS_MASK) - 1) - W;
            A.src = g * GT + ll;
        }
        A.after_first = tot & ((1ull << 40) - 1);
        A.n_events = (unsigned)(tot >> 40);
        publish(&t.gagg[(size_t)ch * n_groups_of(a.n_tiles) + g], A);
    }
}

// The groups of one channel under the range's carry-in: lane = group, 64 groups per step, the carried state uniform.
__device__ __forceinline__ void range_scan(const ScanArgs& a, const int ch)
{
    const ScanArgs& t = a;
    const int lane = threadIdx.x, n_groups = n_groups_of(a.n_tiles);
    p25fe_anchor_t Ain;
    Ain.valid = 0; Ain.s = 0; Ain.hi = Ain.mid = Ain.lo = 0.f; Ain.period_d = SPS; Ain.period_n = 1;
    if (t.anchor_in) Ain = t.anchor_in[ch];
    int cv = Ain.valid != 0 ? 1 : 0, csrc = -1;
    long cs = Ain.s, cfirst = -1;
    unsigned long long ccnt = 0ull, cev = 0ull, cbase = 0ull;
    const GroupAgg* ga = t.gagg + (size_t)ch * n_groups;
    GroupPre* gp = t.gpre + (size_t)ch * n_groups;
    const long range_end = a.abs0 + a.n;
    for (int c0 = 0; c0 < n_groups; c0 += WV) {
        const int gi = c0 + lane;
        const bool act = gi < n_groups;
        GroupAgg A;
        A.first_event = -1; A.last_s = 0; A.after_first = 0ull; A.src = -1; A.n_events = 0u;
        if (act) A = ga[gi];
        const bool has = A.first_event >= 0;
        const unsigned long long evm = __ballot(has);
        const int pl = last_set_below(evm, lane);
        const long ps = shfl_l(A.last_s, pl < 0 ? 0 : pl);
        const int psrc = __shfl(A.src, pl < 0 ? 0 : pl, 64);
        const int v = pl >= 0 ? 1 : cv, src = pl >= 0 ? psrc : csrc;
        const long s = pl >= 0 ? ps : cs;
        const long G0 = a.abs0 + (long)gi * (GT * TS);
        const long G1 = G0 + (long)GT * TS < range_end ? G0 + (long)GT * TS : range_end;
        const unsigned long long carry = (act && v) ? (unsigned long long)count_instants(s, G0, has ? A.first_event + 1 : G1) : 0ull;
        const unsigned long long mine = (act ? carry + A.after_first : 0ull) | ((unsigned long long)A.n_events << 40);
        const unsigned long long incl = wave_incl_sum64(mine, lane);
        const unsigned long long dib0 = ccnt + ((incl - mine) & ((1ull << 40) - 1));
        if (act) {
            GroupPre P;
            P.s = s; P.dibit_off = dib0; P.carry_cnt = carry; P.src = src; P.valid = v;
            P.event_off = (unsigned)(cev + ((incl - mine) >> 40)); P.pad_ = 0u;
            gp[gi] = P;
        }
        const unsigned long long tot = __shfl(incl, WV - 1, 64);
        if (evm) {                                                   // uniform
            const int fl = __builtin_ctzll(evm), ll = 63 - __builtin_clzll(evm);
            if (cfirst < 0) {                                        // the range's first own detection
                cfirst = shfl_l(A.first_event, fl);
                cbase = __shfl(dib0 + carry, fl, 64);
            }
            cv = 1; cs = shfl_l(A.last_s, ll); csrc = __shfl(A.src, ll, 64);
        }
        ccnt += tot & ((1ull << 40) - 1);
        cev += tot >> 40;
    }
    if (lane == 0) {
        p25fe_result_t r;
        r.n_baseband = t.n_baseband;
        r.n_dibits = ccnt;
        r.n_sync = cev;
        p25fe_anchor_t A = Ain;
        if (csrc >= 0) {
            const TileRec rc = a.recs[(size_t)ch * a.n_tiles + csrc];
            A.valid = 1; A.s = rc.last_s; A.hi = rc.hi; A.mid = rc.mid; A.lo = rc.lo;
        }
        A.period_d = SPS; A.period_n = 1;                          // fixed stride
        r.anchor_out = A;
        r.first_event = cfirst;
        r.n_dibits_after_first = cfirst >= 0 ? ccnt - cbase : 0ull;
        r.carry_end = cfirst >= 0 ? cfirst + 1 : -1;
        r.first_seg_end = -1;
        r.flags = 0u; r.reserved = 0u;
        t.result[ch] = r;
    }
}

#ifndef P25FE_JIT
// The top step alone (fixed-stride receiver): a re-scan of group aggregates that are already there under another carry-in
// (p25fe_shard_pass2 with host-resolved anchors), and the record of an EMPTY range (no tile, no K2: the carry-in is handed through).
__global__ __launch_bounds__(WV, 4) void k_range_scan(ScanArgs a) { range_scan(a, (int)blockIdx.x); }
// K3 of the fixed-stride receiver as its own launch of ONE-WAVE workgroups, one per group of GT tiles (grid: groups x channels): the group
// scan, then -- the channel's last group to arrive -- the scan of the groups.  It fits beside a running K1 (no LDS, 51 VGPRs), where the
// 512-thread k_scan of rounds 2 - 5 waited for the drain.  (K2 keeps its tail-less form here: with the scan in K2's own tail, as the
// general receiver has it, every one of 3 750 detection workgroups lives ~4 us longer -- drain + ticket -- and K1 beside them paid 20 us.)
__global__ __launch_bounds__(WV, 4) void k_scan_tiles(ScanArgs a)
{
    const int g = blockIdx.x, ch = blockIdx.y, n_groups = n_groups_of(a.n_tiles);
    group_scan(a, g, ch);
    if (!last_arrival(a.tickets + (size_t)ch * (n_groups + 1) + n_groups, (unsigned)n_groups)) return;
    range_scan(a, ch);
}
#endif

// ------------------------------------------------------------------------------------------
// K4: slicer.  One wave per tile.  A tile is a short list of segments -- [tile start, first own detection] under the
// carry-in anchor, then one per own detection -- and inside a segment the symbol instants are CONSECUTIVE floats of one
// plane: lane-consecutive loads, lane-consecutive byte stores.
// ------------------------------------------------------------------------------------------
// Carry resolution across time shards (BASELINE.json config 5): the same "latest anchor wins" rule as K3, one level
// up.  summaries[r] were produced assuming no carry-in; shard r's carry-in is the anchor_out of the latest earlier
// shard that has an event of its own, and its dibit offset adds the closed-form count of instants that the carry-in
// governs before the shard's first own event.  Shared by the host entry point and the one-thread device kernel.
__host__ __device__ inline void shard_resolve_impl(const p25fe_result_t* summaries, const uint64_t* shard_bb0,
                                                    const uint64_t* shard_bb_n, int n_shards, int symbol_clock,
                                                    p25fe_anchor_t* anchor_in, uint64_t* dibit_offset)
{
    // With the tracking clock (SPEC 3.8b) a shard that owns baseband [lo, hi) processes [lo - L, hi - L), and the first
    // own detection of a shard takes its period from the carry-in, which pass 1 did not know: its share of
    // n_dibits_after_first (counted at the nominal 10 / 1) is replaced by the count under the real clock.
    const bool track = symbol_clock != 0;
    const long L = track ? CLK_L : 0;
    p25fe_anchor_t cur;
    cur.s = 0; cur.hi = cur.mid = cur.lo = 0.f; cur.valid = 0; cur.period_d = SPS; cur.period_n = 1;
    uint64_t off = 0;
    for (int r = 0; r < n_shards; ++r) {
        anchor_in[r] = cur;
        dibit_offset[r] = off;
        const p25fe_result_t& R = summaries[r];
        const long lo = (long)shard_bb0[r] - L, hi = (long)(shard_bb0[r] + shard_bb_n[r]) - L;
        const long pre_hi = R.carry_end >= 0 ? (long)R.carry_end : hi;
        const uint64_t pre = cur.valid ? (uint64_t)clock_count(cur.s, cur.period_d, cur.period_n, lo, pre_hi) : 0;
        uint64_t own = R.first_event >= 0 ? R.n_dibits_after_first : 0;
        const bool tracks = track && (R.flags & P25FE_RES_FIRST_TRACKS_CARRY) && cur.valid;
        if (R.first_event >= 0 && tracks) {
            const long s0 = (long)R.first_event - W;
            int D0, N0;
            clock_period(true, true, cur.s, frac3((unsigned)cur.valid >> 8), s0, frac3(R.reserved), D0, N0);
            own = own - (uint64_t)clock_count(s0, SPS, 1, s0 + W + 1, (long)R.first_seg_end) +
                  (uint64_t)clock_count(s0, D0, N0, s0 + W + 1, (long)R.first_seg_end);
        }
        off += pre + own;
        if (R.carry_end >= 0) {                                   // the shard has an event: the carry changes
            if (R.anchor_out.valid) {
                p25fe_anchor_t nxt = R.anchor_out;
                if (!clock_plausible(nxt.period_d, nxt.period_n)) { nxt.period_d = SPS; nxt.period_n = 1; }
                if (R.flags & P25FE_RES_OUT_PERIOD_FROM_CARRY)
                    clock_period(track, tracks, cur.s, frac3((unsigned)cur.valid >> 8), nxt.s, frac3((unsigned)nxt.valid >> 8), nxt.period_d, nxt.period_n);
                cur = nxt;
            } else {
                cur.valid = 0;
            }
        }
    }
    dibit_offset[n_shards] = off;            // total: shard r holds offset[r + 1] - offset[r] dibits
}

// Pass 2 of a time shard WITHOUT a second scan and without a separate resolve launch (fixed-stride receiver, no lock drops
// inside the shard).  Pass 1's scan ran with no carry-in; the carry-in anchor `cur` (the anchor the nearest earlier shard
// with an event of its own ended on -- shard_resolve_impl's rule) changes exactly two things, both in closed form:
//   * tiles in front of the shard's first own detection (ScanOut.src < 0) are governed by `cur`: their dibit offset is
//     the number of instants of `cur` between the shard's start and the tile's start;
//   * every later tile's offset grows by the number of instants `cur` governs in the shard, [start, first_event].
// Every slicer workgroup derives `cur` from the all-gathered summaries itself (one backward step in practice); the
// workgroup of tile 0 also runs the full combine for the record: anchors, the n_shards + 1 offsets, the final result.
struct ShardFix {
    const p25fe_result_t* summ; // nullable: [n_shards] pass-1 summaries of all shards in time order (device memory)
    const uint64_t* bb0;        // [n_shards] first owned baseband index of each shard
    const uint64_t* bbn;        // [n_shards] owned baseband samples
    int n_shards, rank;
    p25fe_anchor_t* anc_out;    // [n_shards] every shard's carry-in anchor
    uint64_t* off_out;          // [n_shards + 1] every shard's dibit offset in the capture's stream, last = total
    p25fe_result_t* result;     // this shard's final record (what a pass-2 scan would have written)
};

struct SliceArgs {
    Planar pl;
    long n;
    long abs0;
    int n_tiles;
    const ScanOut* outs;
    const TileRec* recs;
    const unsigned long long* tsum;
    const uint16_t* evl;
    const float* evthr;         // [ch][n_tiles][EVTHR_N][3] from K2
    const p25fe_anchor_t* anchor_in;    // nullable, [ch]
    uint8_t* dibits;            // [ch][dibit_stride]
    long dibit_stride;          // row stride AND per-channel capacity: dibits past it are counted (K3) but not stored
    int64_t* sync_pos;          // nullable
    uint64_t* sync_dibit;       // nullable
    long sync_stride;
    uint8_t* dibits2;           // nullable: a second destination of every dibit, same row layout (rank 0 of a time-sharded capture
                                // slices straight into the ordered stream as well: its shard starts at offset 0)
    ShardFix fix;               // fix.summ != nullptr: pass 2 of a time shard, the combine done here (p25fe_shard_pass2_dev)
    const GroupPre* gpre;       // `outs` are k_scan_tiles' group-local records, completed here by the group's carry-in (group_fix)
};

// A tile's group-local carry-in record (group_scan) + its group's carry-in (range_scan) -> the record k_scan would have written.
__device__ __forceinline__ ScanOut group_fix(ScanOut so, const GroupPre& P, const long abs0, const int tile, int& valid, long& s_abs)
{
    if (so.src < 0) {                                               // in front of the group's first own detection: the group's carry-in governs
        valid = P.valid; s_abs = P.s;
        so.src = P.src;
        so.event_off = P.event_off;
        const long G0 = abs0 + (long)(tile / GT) * ((long)GT * TS);
        so.dibit_off = P.dibit_off + (P.valid ? (unsigned long long)count_instants(P.s, G0, abs0 + (long)tile * TS) : 0ull);
    } else {
        so.event_off += P.event_off;
        so.dibit_off += P.dibit_off + P.carry_cnt;
    }
    return so;
}

// so: the tile's carry-in record; u: its packed summary; (valid, s_abs, hi, mid, lo): the anchor in force at its first sample
__device__ __forceinline__ void slice_tile(const SliceArgs& a, const int tile, const int ch, const ScanOut so,
                                           const unsigned long long u, const int valid, const long s_abs, const float hi,
                                           const float mid, const float lo, const EvLanes el)
{
    const int lane = threadIdx.x;
    const float* f = a.pl.f + (size_t)ch * a.pl.f_ch;
    const long t0 = (long)tile * TS;
    const int tn = a.n - t0 < TS ? (int)(a.n - t0) : TS;
    const int n_ev = (int)((u >> (2 * TS_BITS)) & TS_MASK);
    if (!valid && n_ev == 0) return;                                // nothing decided yet: no instants
    // detection k's decision offset: lane k % 64 of the batch in registers (a tile with more than 64 detections reloads, 64 at a time)
    const uint16_t* evl = a.evl + ((size_t)ch * a.n_tiles + tile) * EVCAP;
    unsigned evr = el.ev;
    int batch = 0;
    auto ev_at = [&](int k) -> int {
        if ((k & ~(WV - 1)) != batch) {                             // uniform
            batch = k & ~(WV - 1);
            evr = batch + lane < EVCAP ? (unsigned)evl[batch + lane] : 0u;
        }
        return rdl((int)evr, k & (WV - 1));
    };
    uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride + so.dibit_off;
    uint8_t* out2 = a.dibits2 ? a.dibits2 + (size_t)ch * a.dibit_stride + so.dibit_off : nullptr;
    // room left in this channel's row from the tile's first dibit on (<= 0: the row is full).  A receiver that re-anchors
    // on every sync word follows the transmitter's symbol clock, so a range can hold more than n / 10 dibits (up to
    // n / (W + 1) under dense detections): the count in p25fe_result_t stays exact, the stores stop at the capacity.
    const long room = a.dibit_stride - (long)so.dibit_off;

    // instants m in [m_lo, m_hi) (tile-local) that are congruent to the anchor: first one at m_lo + off
    auto emit = [&](int off, int m_lo, int m_hi, float h, float m, float l, int rank) -> int {
        const int m_first = m_lo + off;
        if (m_first >= m_hi) return 0;
        const int count = (m_hi - 1 - m_first) / SPS + 1;
        const long p = t0 + m_first + PLPAD;
        const long i0 = p / SPS;
        const float* src = f + (i0 >> 5) * PL_BLK + (int)(p - i0 * SPS) * 32;     // block of symbol i0, row of the plane
        const int o0 = (int)(i0 & 31);
        uint8_t* dst = out + rank;
        uint8_t* dst2 = out2 ? out2 + rank : nullptr;               // uniform
        for (int j0 = 0; j0 < count; j0 += 4 * WV) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q * WV + lane;
                const int jj = o0 + j;                              // symbol index relative to the first block
                v[q] = j < count ? src[(jj >> 5) * PL_BLK + (jj & 31)] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q * WV + lane;
                if (j < count && rank + j < room) {
                    const unsigned char db = slice_dibit(v[q], h, m, l);
                    dst[j] = db;
                    if (dst2) dst2[j] = db;
                }
            }
        }
        return count;
    };

    int rank = 0;
    {
        const int m_hi = n_ev ? (int)(u & TS_MASK) : tn;            // first_off + 1: the instant AT the decision index is the old anchor's
        if (valid) {
            const unsigned ph = (unsigned)((a.abs0 + t0 - s_abs) % SPS);     // tile start is past the anchor: positive
            rank += emit((int)((SPS - ph) % (unsigned)SPS), 0, m_hi, hi, mid, lo, 0);
        }
    }
    int e_nx = n_ev ? ev_at(0) : 0;
    for (int k = 0; k < n_ev; ++k) {
        const int ek = e_nx;
        if (k + 1 < n_ev) e_nx = ev_at(k + 1);
        const int m_hi = k + 1 < n_ev ? e_nx + 1 : tn;
        float h, m, l;
        if (k < EVTHR_N) {                                          // K2 left the thresholds of the tile's first detections
            h = rdl_f(el.th, 3 * k); m = rdl_f(el.th, 3 * k + 1); l = rdl_f(el.th, 3 * k + 2);
        } else {
            float v[NSYN];
            sync_gather(f, t0 + ek - W + PLPAD, v);                 // uniform window; same arithmetic as K2, same bits
            sync_thresholds(v, h, m, l);
        }
        if (lane == 0 && a.sync_pos && (long)(so.event_off + k) < a.sync_stride) {
            a.sync_pos[(size_t)ch * a.sync_stride + so.event_off + k] = a.abs0 + t0 + ek - W;
            // index of the first dibit this detection governs = dibits for instants <= e_k
            a.sync_dibit[(size_t)ch * a.sync_stride + so.event_off + k] = so.dibit_off + (unsigned long long)rank;
        }
        // instants n > e_k with n = s_k + 10 j: the first is s_k + 10 = e_k + 5
        rank += emit(SPS - W - 1, ek + 1, m_hi, h, m, l, rank);
    }
}

__device__ __forceinline__ void slice_item(const SliceArgs& a, const int tile, const int ch)
{
    const ShardFix& x = a.fix;
    const bool fixm = x.summ != nullptr;                            // uniform: pass 2 of a time shard (one channel), see ShardFix
    if (fixm && tile == a.n_tiles) {
        // the grid's EXTRA workgroup: the full combine for the record (every shard's carry-in anchor, the n_shards + 1 dibit
        // offsets, this shard's final result) -- beside the slicing workgroups, none of which waits for it
        if (threadIdx.x == 0) {
            shard_resolve_impl(x.summ, x.bb0, x.bbn, x.n_shards, 0, x.anc_out, x.off_out);
            p25fe_result_t r = x.summ[x.rank];
            const p25fe_anchor_t cur = x.anc_out[x.rank];           // (this thread's own stores)
            const long r_lo = a.abs0, r_hi = a.abs0 + a.n;
            const long pre_hi = r.first_event >= 0 ? r.first_event + 1 : r_hi;
            const unsigned long long pre_total = cur.valid ? (unsigned long long)count_instants(cur.s, r_lo, pre_hi) : 0ull;
            r.n_dibits = pre_total + (r.first_event >= 0 ? r.n_dibits_after_first : 0ull);
            if (r.first_event < 0) r.anchor_out = cur;              // no detection of its own: the shard ends on what it started with
            r.anchor_out.period_d = SPS; r.anchor_out.period_n = 1;
            *x.result = r;
        }
        return;
    }
    // the shard's own summary and its predecessor's are requested together with the tile's records: one memory latency
    long sh_first_event = -1;
    p25fe_result_t prev;
    prev.carry_end = -1; prev.anchor_out.valid = 0;
    if (fixm) {
        sh_first_event = x.summ[x.rank].first_event;
        if (x.rank > 0) prev = x.summ[x.rank - 1];
    }
    ScanOut so = a.outs[(size_t)ch * a.n_tiles + tile];
    const unsigned long long u = a.tsum[(size_t)ch * a.n_tiles + tile];
    const EvLanes el = ev_lanes_load(a.evl + ((size_t)ch * a.n_tiles + tile) * EVCAP, nullptr, a.evthr + ((size_t)ch * a.n_tiles + tile) * (EVTHR_N * 3));
    // carry-in anchor
    int valid = 0;
    long s_abs = 0;
    float hi = 0.f, mid = 0.f, lo = 0.f;
    {
        const GroupPre P = a.gpre[(size_t)ch * n_groups_of(a.n_tiles) + tile / GT];
        so = group_fix(so, P, a.abs0, tile, valid, s_abs);
    }
    if (so.src >= 0) {
        const TileRec t = a.recs[(size_t)ch * a.n_tiles + so.src];
        valid = 1; s_abs = t.last_s; hi = t.hi; mid = t.mid; lo = t.lo;
    } else if (a.anchor_in) {
        const p25fe_anchor_t A = a.anchor_in[ch];
        valid = A.valid; s_abs = A.s; hi = A.hi; mid = A.mid; lo = A.lo;
    }
    if (fixm) {
        // the anchor the nearest earlier shard with an event of its own ended on (normally the one right before)
        p25fe_anchor_t cur;
        cur.s = 0; cur.hi = cur.mid = cur.lo = 0.f; cur.valid = 0; cur.period_d = SPS; cur.period_n = 1;
        for (int r = x.rank - 1; r >= 0; --r) {
            const p25fe_result_t R = r == x.rank - 1 ? prev : x.summ[r];
            if (R.carry_end >= 0) {
                if (R.anchor_out.valid) cur = R.anchor_out;
                break;
            }
        }
        const long r_lo = a.abs0, r_hi = a.abs0 + a.n;
        if (so.src < 0) {
            valid = cur.valid != 0; s_abs = cur.s; hi = cur.hi; mid = cur.mid; lo = cur.lo;
            so.dibit_off = valid ? (unsigned long long)count_instants(cur.s, r_lo, r_lo + (long)tile * TS) : 0ull;
        } else if (cur.valid) {
            // the instant AT the first decision index is still the carry-in's
            so.dibit_off += (unsigned long long)count_instants(cur.s, r_lo, sh_first_event >= 0 ? sh_first_event + 1 : r_hi);
        }
    }
    slice_tile(a, tile, ch, so, u, valid, s_abs, hi, mid, lo, el);
}
#ifndef P25FE_JIT
// (One workgroup per tile.  A BOUNDED grid of persistent workgroups walking the tiles -- 1 024, i.e. 4 per CU, which leave K1 beside
// them its 10 workgroups per CU -- was built and measured in round 6: a 400-step pipelined run cost the same as this form, the chain
// took ~240 us instead of ~60, so that the driver's 20-step command, whose LAST chain overlaps nothing, lost 1.5 - 4 %, and the loop
// cost 25 - 70 VGPRs per kernel (SGPR spills).  profiles/r06_rx_grid_ab.txt; not adopted.)
__global__ __launch_bounds__(WV, 4) void k_slice(SliceArgs a) { slice_item(a, (int)blockIdx.x, (int)blockIdx.y); }
#endif

// ------------------------------------------------------------------------------------------
// K3 / K4 of the general receiver (tracking clock and / or lock drops inside the range).
//
// The receiver's state at a tile's first sample is the state after the latest tile that has an event (a detection or a
// lock drop).  That state is: locked or not; if locked, the last detection's position and clock.  The clock of a tile's
// LAST detection is known inside the tile unless that detection is the tile's only event and tracks the carry-in --
// then it is the interval from the previous detection, i.e. from the event tile BEFORE: the state depends on the two
// latest event tiles, never on more.  So one "latest two" scan gives every tile its carry-in, each tile's dibit count
// follows independently in closed form (clock_count: two 64-bit divisions per anchor and interval), and a prefix sum
// gives the offsets.
// ------------------------------------------------------------------------------------------
struct ScanOutG {               // per (channel, tile) carry-in written by k_scan_g_groups
    long s;                     // position of the detection in force at the tile's first sample
    unsigned long long dibit_off;
    int src;                    // tile whose record holds that detection's thresholds; -1: the range's anchor_in; -2: not locked
    unsigned event_off;
    int D, N;                   // its clock
    int f;                      // quarter-sample fraction of its sync position (enters the NEXT detection's period only)
    int pad_;
};

struct CState { int valid; long s; int D, N; int src; int f; };

// ------------------------------------------------------------------------------------------
// The general receiver's scan, hierarchical like the fixed-stride one (ScanArgs above) -- the 512-thread k_scan_g of rounds 3 - 5
// (48 - 53 KB of LDS) is gone: beside a running K1 it waited for the drain, and the tracking clock's pipelined step paid for it.
//   pass A   k_scan_tiles_g, one wave per group of GT tiles: the group is scanned with NO carry-in (scan_g_lanes: lane =
//            tile, the "latest two event tiles" rule by two bit scans of a ballot, neighbours' fields by lane shuffles) and leaves
//            a summary of what a carry-in would change -- the same fields a time shard's pass 1 leaves in p25fe_result_t, for the
//            same reason (GroupSumG);
//   top      the channel's last group to arrive walks the summaries under the range's carry-in (range_scan_g: shard_resolve_impl's rule, one
//            level down): every group's carry-in state and offsets (GroupPreG), and the range's record;
//   pass B   k_scan_g_groups, one wave per group: the same lane scan under the group's real carry-in -> the per-tile carry-ins
//            (ScanOutG) that k_slice_g / k_ev_collect read, as k_scan_g wrote them.
// A re-scan under another carry-in (a time shard's pass 2) is k_range_scan_g (the top step alone, on pass 1's summaries) + pass B.
// ------------------------------------------------------------------------------------------
struct GroupSumG {              // per (channel, group): the group's tiles under NO carry-in
    long first_event;           // decision index of the group's first detection, -1: none
    long carry_end;             // index of the group's first event of any kind (GS_HAS_EVENT)
    long first_seg_end;         // where the first detection's governed interval ends inside the group (the group's end if GS_SEG_OPEN)
    unsigned long long after_first;     // instants the group's own detections govern, the first one's interval counted at 10 / 1
    long out_s;                 // the state the group ends in, if it has an event: position, fraction, tile, clock of the last detection
    int out_valid, out_f, out_src, out_D, out_N;
    unsigned n_sync;
    unsigned flags;             // GS_*
    int first_f;                // the first detection's fraction
    int n_event_tiles;
    int pad_;
};
constexpr unsigned GS_FIRST_TRACKS = 1u;      // no lock drop between the group's start and its first detection (P25FE_RES_FIRST_TRACKS_CARRY)
constexpr unsigned GS_OUT_FROM_CARRY = 2u;    // out_D / out_N are the interval from the carry-in to the group's only detection (P25FE_RES_OUT_PERIOD_FROM_CARRY)
constexpr unsigned GS_HAS_EVENT = 4u;
constexpr unsigned GS_SEG_OPEN = 8u;          // the first detection's interval is still open at the group's end
struct GroupPreG {              // per (channel, group): the state at its first sample and what lies in front of it
    long s;
    unsigned long long dibit_off;
    int valid, D, N, src, f;
    unsigned event_off;
};

struct GLanes {
    CState st;                  // the receiver's state at my tile's first sample
    unsigned pre;               // instants of my tile under it
    unsigned long long excl;    // dibits (low 40 bits) | detections << 40 of the group in front of my tile
    unsigned long long total;   // ... of the whole group
    unsigned long long evm;     // the group's tiles with an event of any kind
};
// the state after event tile (lane) t1, t2 = the event tile before it, -1: the carry-in C.  Every lane calls it (shuffles).
__device__ __forceinline__ CState state_after_lane(const int t1, const int t2, const CState& C, const bool track, const TileSumG& g,
                                                   const long last_s, const int last_f, const int tile0)
{
    const int i1 = t1 < 0 ? 0 : t1, i2 = t2 < 0 ? 0 : t2;
    const unsigned fl1 = (unsigned)__shfl((int)g.n_det_flags, i1, 64) >> 16, fl2 = (unsigned)__shfl((int)g.n_det_flags, i2, 64) >> 16;
    const int D1 = __shfl(g.out_D, i1, 64), N1 = __shfl(g.out_N, i1, 64);
    const long s1 = shfl_l(last_s, i1), s2 = shfl_l(last_s, i2);
    const int f1 = __shfl(last_f, i1, 64), f2 = __shfl(last_f, i2, 64);
    if (t1 < 0) return C;
    CState st;
    if (!(fl1 & G_OUT_VALID)) { st.valid = 0; st.s = 0; st.D = SPS; st.N = 1; st.src = -2; st.f = 0; return st; }
    st.valid = 1; st.src = tile0 + t1; st.s = s1; st.f = f1;
    if (fl1 & G_OUT_PERIOD_KNOWN) { st.D = D1; st.N = N1; return st; }
    bool pv; long ps; int pf;
    if (t2 >= 0) { pv = (fl2 & G_OUT_VALID) != 0; ps = pv ? s2 : 0; pf = pv ? f2 : 0; }
    else { pv = C.valid != 0; ps = C.s; pf = C.f; }                // (what the state after ANY earlier event tile hands on: locked?, position, fraction)
    clock_period(track, pv, ps, pf, st.s, st.f, st.D, st.N);
    return st;
}
// One group under the carry-in C: lane = tile tile0 + lane (g all zero for a lane past the range), T0 / tn my tile's first index / length.
__device__ __forceinline__ GLanes scan_g_lanes(const TileSumG& g, const long last_s, const int last_f, const CState& C, const bool track,
                                               const long T0, const int tn, const int tile0)
{
    GLanes o;
    const int lane = threadIdx.x;
    o.evm = __ballot(g.pre_end1 != 0u);
    const int t1 = last_set_below(o.evm, lane);
    const int t2 = t1 >= 0 ? last_set_below(o.evm, t1) : -1;
    o.st = state_after_lane(t1, t2, C, track, g, last_s, last_f, tile0);
    const long pre_hi = g.pre_end1 ? T0 + (long)g.pre_end1 - 1 : T0 + tn;
    o.pre = (tn > 0 && o.st.valid) ? (unsigned)clock_count(o.st.s, o.st.D, o.st.N, T0, pre_hi) : 0u;
    unsigned tot = o.pre;
    if (g.first1) {
        const long s0 = T0 + (long)(g.first1 & 0xffffu) - 1 - W;
        int D0, N0;
        clock_period(track, ((g.n_det_flags >> 16) & G_FIRST_TRACKS) && o.st.valid, o.st.s, o.st.f, s0, frac3(g.first1 >> 16), D0, N0);
        tot += (unsigned)clock_count(s0, D0, N0, s0 + W + 1, T0 + g.end0) + g.post_rest;
    }
    const unsigned long long mine = (unsigned long long)tot | ((unsigned long long)(g.n_det_flags & 0xffffu) << 40);
    const unsigned long long incl = wave_incl_sum64(mine, lane);
    o.excl = incl - mine;
    o.total = __shfl(incl, WV - 1, 64);
    return o;
}

struct ScanArgsG {
    const TileSumG* gsum;
    const TileRec* recs;
    ScanOutG* outs;
    GroupSumG* gsg;                     // [ch][n_groups]
    GroupPreG* gpg;                     // [ch][n_groups]
    unsigned* tickets;                  // [ch][n_groups + 1], zero between launches
    int n_tiles;
    long n;
    long abs0;                          // absolute index of the first PROCESSED sample (owned sample 0 minus the lookahead)
    const p25fe_anchor_t* anchor_in;    // nullable, [ch]
    p25fe_result_t* result;             // [ch]
    unsigned long long n_baseband;
    int track;
};

// my lane's tile of group g: summary, last detection, geometry
__device__ __forceinline__ void load_lane_tile(const ScanArgsG& a, const int g, const int ch, TileSumG& gs, long& last_s, int& last_f, long& T0, int& tn)
{
    const int tl = g * GT + (int)threadIdx.x;
    gs.pre_end1 = 0u; gs.first1 = 0u; gs.end0 = 0u; gs.last1 = 0u; gs.n_det_flags = 0u; gs.post_rest = 0u; gs.out_D = SPS; gs.out_N = 1;
    last_s = 0; last_f = 0; tn = 0;
    T0 = a.abs0 + (long)tl * TS;
    if (tl < a.n_tiles) {
        gs = a.gsum[(size_t)ch * a.n_tiles + tl];
        const TileRec* rc = a.recs + (size_t)ch * a.n_tiles + tl;
        last_s = rc->last_s; last_f = rc->last_f;
        const long rem = a.n - (long)tl * TS;
        tn = rem < TS ? (int)rem : TS;
    }
}

// pass A: group g with no carry-in -> its summary
__device__ __forceinline__ void group_scan_g(const ScanArgsG& a, const int g, const int ch)
{
    const int lane = threadIdx.x, tile0 = g * GT;
    TileSumG gs; long last_s, T0; int last_f, tn;
    load_lane_tile(a, g, ch, gs, last_s, last_f, T0, tn);
    CState C;
    C.valid = 0; C.s = 0; C.D = SPS; C.N = 1; C.src = -2; C.f = 0;
    const bool track = a.track != 0;
    const GLanes L = scan_g_lanes(gs, last_s, last_f, C, track, T0, tn, tile0);
    // the state the group ends in: after its latest event tile
    const int ta = L.evm ? 63 - __builtin_clzll(L.evm) : -1, tb = ta >= 0 ? last_set_below(L.evm, ta) : -1;
    const CState se = state_after_lane(ta, tb, C, track, gs, last_s, last_f, tile0);
    const unsigned fl_a = (unsigned)__shfl((int)gs.n_det_flags, ta < 0 ? 0 : ta, 64) >> 16;
    // the first detection
    const unsigned long long dm = __ballot(gs.first1 != 0u);
    const int fd = dm ? __builtin_ctzll(dm) : 0;
    const unsigned long long after = L.evm & ~((2ull << fd) - 1ull);  // event tiles behind the first detection's
    const int nx = after ? __builtin_ctzll(after) : 0;
    const int e1 = L.evm ? __builtin_ctzll(L.evm) : 0;
    // (uniform values, read from the lanes that hold them)
    const unsigned first1 = (unsigned)__shfl((int)gs.first1, fd, 64), end0 = (unsigned)__shfl((int)gs.end0, fd, 64);
    const unsigned flg_fd = (unsigned)__shfl((int)gs.n_det_flags, fd, 64) >> 16;
    const long T0_fd = shfl_l(T0, fd);
    const int tn_fd = __shfl(tn, fd, 64);
    const unsigned long long base_first = __shfl(L.excl + (unsigned long long)L.pre, fd, 64) & ((1ull << 40) - 1);
    const long nx_first = shfl_l(T0 + (long)gs.pre_end1 - 1, nx);
    const long carry_end = shfl_l(T0 + (long)gs.pre_end1 - 1, e1);
    if (lane == 0) {
        GroupSumG S;
        S.first_event = -1; S.carry_end = -1; S.first_seg_end = -1; S.after_first = 0ull; S.first_f = 0;
        S.out_s = se.s; S.out_valid = se.valid; S.out_f = se.f; S.out_src = se.src; S.out_D = se.D; S.out_N = se.N;
        S.n_sync = (unsigned)(L.total >> 40);
        S.n_event_tiles = __popcll(L.evm);
        S.pad_ = 0;
        unsigned fl = 0u;
        if (L.evm) { fl |= GS_HAS_EVENT; S.carry_end = carry_end; }
        if (dm) {
            S.first_event = T0_fd + (long)(first1 & 0xffffu) - 1;
            S.first_f = frac3(first1 >> 16);
            S.after_first = (L.total & ((1ull << 40) - 1)) - base_first;
            if ((flg_fd & G_FIRST_TRACKS) && e1 == fd) fl |= GS_FIRST_TRACKS;      // (and no event tile in front of it)
            long fse = T0_fd + (long)end0;
            if (fse >= T0_fd + tn_fd) {                               // still open at its tile's end: runs on to the next event tile, or out of the group
                if (after) fse = nx_first;
                else { fse = a.abs0 + ((long)(tile0 + GT) * TS < a.n ? (long)(tile0 + GT) * TS : a.n); fl |= GS_SEG_OPEN; }
            }
            S.first_seg_end = fse;
        }
        // the clock the group ends on was taken from the carry-in: its only event tile holds one tracking detection
        if (se.src >= 0 && tb < 0 && !(fl_a & G_OUT_PERIOD_KNOWN)) fl |= GS_OUT_FROM_CARRY;
        S.flags = fl;
        publish(&a.gsg[(size_t)ch * n_groups_of(a.n_tiles) + g], S);
    }
}

// top: the groups of one channel under the range's carry-in (every lane runs the same walk; 64 summaries are fetched per step, one per
// lane, and read lane by lane).  shard_resolve_impl's rule with the range's carry-in as the starting state.
__device__ __forceinline__ void range_scan_g(const ScanArgsG& a, const int ch)
{
    const int lane = threadIdx.x, n_groups = n_groups_of(a.n_tiles);
    const bool track = a.track != 0;
    p25fe_anchor_t Ain;
    Ain.valid = 0; Ain.s = 0; Ain.hi = Ain.mid = Ain.lo = 0.f; Ain.period_d = SPS; Ain.period_n = 1;
    if (a.anchor_in) Ain = a.anchor_in[ch];
    if (!clock_plausible(Ain.period_d, Ain.period_n)) { Ain.period_d = SPS; Ain.period_n = 1; }
    CState cur;
    cur.valid = Ain.valid != 0; cur.s = Ain.s; cur.D = Ain.period_d; cur.N = Ain.period_n; cur.src = Ain.valid ? -1 : -2;
    cur.f = frac3((unsigned)Ain.valid >> 8);                         // the carry-in's fraction rides in bits 8..10 of `valid`
    unsigned long long off = 0ull, ev = 0ull, base_first = 0ull;
    long first_event = -1, carry_end = 0, fse = -1;
    int first_f = 0, n_event_tiles = 0;
    bool has_event = false, fse_open = false, last_from_carry = false;
    unsigned fl_first = 0u;
    const GroupSumG* gsg = a.gsg + (size_t)ch * n_groups;
    GroupPreG* gpg = a.gpg + (size_t)ch * n_groups;
    const long range_end = a.abs0 + a.n;
    for (int c0 = 0; c0 < n_groups; c0 += WV) {
        GroupSumG M;
        M.first_event = -1; M.carry_end = -1; M.first_seg_end = -1; M.after_first = 0ull; M.out_s = 0; M.out_valid = 0; M.out_f = 0; M.out_src = -2;
        M.out_D = SPS; M.out_N = 1; M.n_sync = 0u; M.flags = 0u; M.first_f = 0; M.n_event_tiles = 0; M.pad_ = 0;
        if (c0 + lane < n_groups) M = gsg[c0 + lane];
        const int cn = n_groups - c0 < WV ? n_groups - c0 : WV;
        GroupPreG mine;                                             // the carry-in of group c0 + lane, kept by the lane that stores it
        mine.s = 0; mine.dibit_off = 0ull; mine.valid = 0; mine.D = SPS; mine.N = 1; mine.src = -2; mine.f = 0; mine.event_off = 0u;
        // The common case, in parallel (lane = group): when every group of the step but its last has an event and ends in a state that
        // does not depend on its own carry-in, the state a group STARTS in is simply what its left neighbour's summary hands on, and
        // the 64 groups' counts (two to eight 64-bit divisions each under a tracked clock) are independent.  Anything else -- an
        // event-free group, a group whose only detection takes its period from the carry-in -- takes the walk below.  Same numbers.
        {
            const bool act = c0 + lane < n_groups;
            const bool evt = (M.flags & GS_HAS_EVENT) != 0u;
            const int hands_on = (evt && !(M.flags & GS_OUT_FROM_CARRY)) ? 1 : 0;
            const int left_ok = __shfl_up(hands_on, 1, 64);
            if (__all(!act || lane == 0 || left_ok != 0)) {              // uniform
                CState S;                                               // the state my group starts in
                S.valid = __shfl_up(M.out_valid, 1, 64); S.s = (long)__shfl_up((unsigned long long)M.out_s, 1, 64);
                S.D = __shfl_up(M.out_D, 1, 64); S.N = __shfl_up(M.out_N, 1, 64); S.src = __shfl_up(M.out_src, 1, 64); S.f = __shfl_up(M.out_f, 1, 64);
                if (!S.valid) { S.s = 0; S.D = SPS; S.N = 1; S.src = -2; S.f = 0; }
                if (lane == 0) S = cur;
                const long G0 = a.abs0 + (long)(c0 + lane) * ((long)GT * TS);
                const long G1 = G0 + (long)GT * TS < range_end ? G0 + (long)GT * TS : range_end;
                unsigned long long pre = 0ull, own = 0ull;
                if (act) {
                    pre = S.valid ? (unsigned long long)clock_count(S.s, S.D, S.N, G0, evt ? M.carry_end : G1) : 0ull;
                    own = M.first_event >= 0 ? M.after_first : 0ull;
                    if (M.first_event >= 0 && track && (M.flags & GS_FIRST_TRACKS) && S.valid) {
                        const long s0 = M.first_event - W;
                        int D0, N0;
                        clock_period(true, true, S.s, S.f, s0, M.first_f, D0, N0);
                        own = own - (unsigned long long)clock_count(s0, SPS, 1, s0 + W + 1, M.first_seg_end) +
                              (unsigned long long)clock_count(s0, D0, N0, s0 + W + 1, M.first_seg_end);
                    }
                }
                const unsigned long long tot = pre + own, ie = wave_incl_sum64(act ? (unsigned long long)M.n_sync : 0ull, lane);
                const unsigned long long it = wave_incl_sum64(tot, lane);
                if (act) {
                    mine.s = S.s; mine.dibit_off = off + (it - tot); mine.valid = S.valid; mine.D = S.D; mine.N = S.N; mine.src = S.src; mine.f = S.f;
                    mine.event_off = (unsigned)(ev + (ie - (unsigned long long)M.n_sync));
                    gpg[c0 + lane] = mine;
                }
                // the range's bookkeeping, in the walk's order
                const unsigned long long evm = __ballot(act && evt), fdm = __ballot(act && M.first_event >= 0);
                if (evm && !has_event) { has_event = true; carry_end = rdl_l(M.carry_end, __builtin_ctzll(evm)); }
                if (evm && first_event >= 0 && fse_open) { fse = rdl_l(M.carry_end, __builtin_ctzll(evm)); fse_open = false; }
                if (fdm && first_event < 0) {
                    const int lf = __builtin_ctzll(fdm);
                    first_event = rdl_l(M.first_event, lf); first_f = rdl(M.first_f, lf);
                    base_first = off + (unsigned long long)rdl_l((long)(it - tot + pre), lf);
                    fl_first = ((rdl((int)M.flags, lf) & (int)GS_FIRST_TRACKS) && n_event_tiles == 0 && !(evm & ((1ull << lf) - 1ull))) ? 1u : 0u;
                    fse = rdl_l(M.first_seg_end, lf); fse_open = (rdl((int)M.flags, lf) & (int)GS_SEG_OPEN) != 0;
                    const unsigned long long later = evm & ~((2ull << lf) - 1ull);
                    if (fse_open && later) { fse = rdl_l(M.carry_end, __builtin_ctzll(later)); fse_open = false; }
                }
                n_event_tiles += wave_sum_i(act ? M.n_event_tiles : 0);
                if (evm) last_from_carry = (rdl((int)M.flags, 63 - __builtin_clzll(evm)) & (int)GS_OUT_FROM_CARRY) != 0;
                // the state after the step's last group
                CState A = S;
                if (act && evt) {
                    if (M.out_valid) {
                        A.valid = 1; A.s = M.out_s; A.D = M.out_D; A.N = M.out_N; A.src = M.out_src; A.f = M.out_f;
                        if (M.flags & GS_OUT_FROM_CARRY) clock_period(track, S.valid != 0, S.s, S.f, A.s, A.f, A.D, A.N);
                    } else {
                        A.valid = 0; A.s = 0; A.D = SPS; A.N = 1; A.src = -2; A.f = 0;
                    }
                }
                cur.valid = rdl(A.valid, cn - 1); cur.s = rdl_l(A.s, cn - 1); cur.D = rdl(A.D, cn - 1); cur.N = rdl(A.N, cn - 1);
                cur.src = rdl(A.src, cn - 1); cur.f = rdl(A.f, cn - 1);
                off += (unsigned long long)rdl_l((long)it, WV - 1);
                ev += (unsigned long long)rdl_l((long)ie, WV - 1);
                continue;
            }
        }
        for (int j = 0; j < cn; ++j) {                               // uniform
            GroupSumG R;
            R.first_event = rdl_l(M.first_event, j); R.carry_end = rdl_l(M.carry_end, j); R.first_seg_end = rdl_l(M.first_seg_end, j);
            R.after_first = (unsigned long long)rdl_l((long)M.after_first, j); R.out_s = rdl_l(M.out_s, j);
            R.out_valid = rdl(M.out_valid, j); R.out_f = rdl(M.out_f, j); R.out_src = rdl(M.out_src, j);
            R.out_D = rdl(M.out_D, j); R.out_N = rdl(M.out_N, j);
            R.n_sync = (unsigned)rdl((int)M.n_sync, j); R.flags = (unsigned)rdl((int)M.flags, j);
            R.first_f = rdl(M.first_f, j); R.n_event_tiles = rdl(M.n_event_tiles, j);
            if (lane == j) {
                mine.s = cur.s; mine.dibit_off = off; mine.valid = cur.valid; mine.D = cur.D; mine.N = cur.N; mine.src = cur.src; mine.f = cur.f;
                mine.event_off = (unsigned)ev;
            }
            const long G0 = a.abs0 + (long)(c0 + j) * ((long)GT * TS);
            const long G1 = G0 + (long)GT * TS < range_end ? G0 + (long)GT * TS : range_end;
            const bool evt = (R.flags & GS_HAS_EVENT) != 0u;
            const unsigned long long pre = cur.valid ? (unsigned long long)clock_count(cur.s, cur.D, cur.N, G0, evt ? R.carry_end : G1) : 0ull;
            unsigned long long own = R.first_event >= 0 ? R.after_first : 0ull;
            const bool tracks = track && (R.flags & GS_FIRST_TRACKS) && cur.valid;
            if (R.first_event >= 0 && tracks) {
                const long s0 = R.first_event - W;
                int D0, N0;
                clock_period(true, true, cur.s, cur.f, s0, R.first_f, D0, N0);
                own = own - (unsigned long long)clock_count(s0, SPS, 1, s0 + W + 1, R.first_seg_end) +
                      (unsigned long long)clock_count(s0, D0, N0, s0 + W + 1, R.first_seg_end);
            }
            if (evt && !has_event) { has_event = true; carry_end = R.carry_end; }
            if (evt && first_event >= 0 && fse_open) { fse = R.carry_end; fse_open = false; }    // the event that ends the first detection's open interval
            if (R.first_event >= 0 && first_event < 0) {
                first_event = R.first_event; first_f = R.first_f;
                base_first = off + pre;
                fl_first = ((R.flags & GS_FIRST_TRACKS) && n_event_tiles == 0) ? 1u : 0u;
                fse = R.first_seg_end; fse_open = (R.flags & GS_SEG_OPEN) != 0u;
            }
            off += pre + own;
            ev += R.n_sync;
            if (evt) {
                n_event_tiles += R.n_event_tiles;
                last_from_carry = (R.flags & GS_OUT_FROM_CARRY) != 0u;
                if (R.out_valid) {
                    CState nxt;
                    nxt.valid = 1; nxt.s = R.out_s; nxt.D = R.out_D; nxt.N = R.out_N; nxt.src = R.out_src; nxt.f = R.out_f;
                    if (R.flags & GS_OUT_FROM_CARRY) clock_period(track, cur.valid != 0, cur.s, cur.f, nxt.s, nxt.f, nxt.D, nxt.N);
                    cur = nxt;
                } else {
                    cur.valid = 0; cur.s = 0; cur.D = SPS; cur.N = 1; cur.src = -2; cur.f = 0;
                }
            }
        }
        if (c0 + lane < n_groups) gpg[c0 + lane] = mine;
    }
    if (lane == 0) {
        p25fe_result_t r;
        r.n_baseband = a.n_baseband;
        r.n_dibits = off;
        r.n_sync = ev;
        p25fe_anchor_t A = Ain;
        if (cur.src >= 0) {
            const TileRec t = a.recs[(size_t)ch * a.n_tiles + cur.src];
            // bit 0: locked; bits 8..10: the sync position's fraction (SPEC 3.8b) -- only when the clock tracks: with the fixed
            // stride the general receiver (a lock-drop list) hands on the same `valid = 1` as the one-tile fast path
            A.valid = track ? (1 | ((cur.f & 7) << 8)) : 1;
            A.s = t.last_s; A.hi = t.hi; A.mid = t.mid; A.lo = t.lo; A.period_d = cur.D; A.period_n = cur.N;
        } else if (cur.src == -2) {
            A.valid = 0;
        }
        r.anchor_out = A;
        r.first_event = first_event;
        r.n_dibits_after_first = first_event >= 0 ? off - base_first : 0ull;
        // -1 means "no event": an event IN FRONT of index 0 (a lock drop at sample 0 or 1 of a fresh stream, which the tracking
        // clock's lookahead puts at -2 / -1; nothing can be locked there) is reported at 0
        r.carry_end = has_event ? (carry_end > 0 ? carry_end : 0) : -1;
        // where the first detection's governed interval ends IN THE RANGE: at the next event, or -- still open -- at the range's end
        // (p25fe_shard_resolve recounts that interval under the detection's real clock)
        r.first_seg_end = first_event >= 0 ? (fse_open ? range_end : fse) : -1;
        unsigned fl = fl_first;
        // the clock the range ends on was taken from the carry-in: its only event is one tracking detection
        if (cur.src >= 0 && n_event_tiles == 1 && last_from_carry) fl |= 2u;
        r.flags = fl; r.reserved = first_event >= 0 ? (unsigned)(first_f & 7) : 0u;     // the first own detection's fraction
        a.result[ch] = r;
    }
}

#ifndef P25FE_JIT
// pass B: one wave per group, the lane scan under the group's carry-in -> the tiles' carry-ins
__global__ __launch_bounds__(WV, 4) void k_scan_g_groups(ScanArgsG a)
{
    const int g = blockIdx.x, ch = blockIdx.y, tile0 = g * GT;
    TileSumG gs; long last_s, T0; int last_f, tn;
    load_lane_tile(a, g, ch, gs, last_s, last_f, T0, tn);
    const GroupPreG P = a.gpg[(size_t)ch * n_groups_of(a.n_tiles) + g];
    CState C;
    C.valid = P.valid; C.s = P.s; C.D = P.D; C.N = P.N; C.src = P.src; C.f = P.f;
    const GLanes L = scan_g_lanes(gs, last_s, last_f, C, a.track != 0, T0, tn, tile0);
    const int tl = tile0 + (int)threadIdx.x;
    if (tl < a.n_tiles) {
        ScanOutG o;
        o.s = L.st.s; o.dibit_off = P.dibit_off + (L.excl & ((1ull << 40) - 1)); o.src = L.st.src;
        o.event_off = P.event_off + (unsigned)(L.excl >> 40); o.D = L.st.D; o.N = L.st.N; o.f = L.st.f; o.pad_ = 0;
        a.outs[(size_t)ch * a.n_tiles + tl] = o;
    }
}
// the top step alone: a re-scan of summaries that are already there under another carry-in (p25fe_shard_pass2 and its device form),
// and the record of an EMPTY range (no tile, no K2: the carry-in is handed through)
__global__ __launch_bounds__(WV, 4) void k_range_scan_g(ScanArgsG a) { range_scan_g(a, (int)blockIdx.x); }
// pass A + the top step as a launch of their own (one wave per group; the channel's last group to arrive walks the groups)
__global__ __launch_bounds__(WV, 4) void k_scan_tiles_g(ScanArgsG a)
{
    const int g = blockIdx.x, ch = blockIdx.y, n_groups = n_groups_of(a.n_tiles);
    group_scan_g(a, g, ch);
    if (!last_arrival(a.tickets + (size_t)ch * (n_groups + 1) + n_groups, (unsigned)n_groups)) return;
    range_scan_g(a, ch);
}
#endif

// K2's kernel (7.1 KB of LDS per one-wave workgroup; GEN: 8.4 KB with the fraction table)
template <bool GEN> __device__ __forceinline__ void detect_item(const DetArgs& a, const int tile, const int ch)
{
    if (a.head_flag && tile <= a.head_tile_max) {                   // uniform
        // every head workgroup released its stores (agent scope) before it took its ticket; the acquire after seeing the
        // flag keeps this wave from reading lines its caches held before that
        // The wait is BOUNDED: the head sits behind the halo's ncclRecv on another stream, and a peer that stalls or dies
        // would otherwise leave these workgroups spinning for ever with no way to abort the queue.  After HEAD_WAIT_TICKS of the 100 MHz
        // wall clock the workgroup gives up, says so in head_err and carries on with whatever the planes hold: the step's results are
        // then garbage, and p25fe_shard_head_check / p25fe_shard_offsets report P25FE_ERR_TIMEOUT instead of handing them out.
        const unsigned long long t0 = wall_clock64();
        bool gave_up = false;
        while ((int)(__hip_atomic_load(a.head_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.head_seq) < 0) {
            if (wall_clock64() - t0 > HEAD_WAIT_TICKS) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (gave_up && threadIdx.x == 0) __hip_atomic_store(a.head_err, a.head_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    detect_tile<GEN>(a, tile, ch);
}
template <bool GEN> __global__ __launch_bounds__(WV, GEN ? 2 : 4) void k_detect(DetArgs a) { detect_item<GEN>(a, (int)blockIdx.x, (int)blockIdx.y); }

struct SliceArgsG {
    Planar pl;
    long n;
    long abs0;
    int n_tiles;
    const ScanOutG* outs;
    const TileSumG* gsum;
    const TileRec* recs;
    const uint16_t* evl;
    const uint32_t* evg;
    const float* evthr;
    const p25fe_anchor_t* anchor_in;
    uint8_t* dibits;
    long dibit_stride;
    int64_t* sync_pos;
    uint64_t* sync_dibit;
    long sync_stride;
    int track;
    uint8_t* dibits2;           // nullable: second destination, as in SliceArgs
};

#ifndef P25FE_JIT
__device__ __forceinline__ void slice_g_item(const SliceArgsG& a, const int tile, const int ch)
{
    __shared__ float CI[P25FE_CLK_PHASES * 4];
    const int lane = threadIdx.x;
    const float* f = a.pl.f + (size_t)ch * a.pl.f_ch;
    const long t0 = (long)tile * TS;
    const int tn = a.n - t0 < TS ? (int)(a.n - t0) : TS;
    const size_t ti = (size_t)ch * a.n_tiles + tile;
    const ScanOutG so = a.outs[ti];
    const TileSumG g = a.gsum[ti];
    const uint16_t* evl = a.evl + ti * EVCAP;
    const uint32_t* evg = a.evg + ti * EVCAP;
    const EvLanes el = ev_lanes_load(evl, evg, a.evthr + ti * (EVTHR_N * 3));      // (with the tile's records: one round trip)
    const int n_ev = (int)(g.n_det_flags & 0xffffu);
    const bool track = a.track != 0;

    int valid = 0;
    float hi = 0.f, mid = 0.f, lo = 0.f;
    if (so.src >= 0) {
        const TileRec t = a.recs[(size_t)ch * a.n_tiles + so.src];
        valid = 1; hi = t.hi; mid = t.mid; lo = t.lo;
    } else if (so.src == -1 && a.anchor_in) {
        const p25fe_anchor_t A = a.anchor_in[ch];
        valid = A.valid; hi = A.hi; mid = A.mid; lo = A.lo;
    }
    if (!valid && n_ev == 0) return;
    if (track) {
        for (int k = lane; k < P25FE_CLK_PHASES * 4; k += WV) CI[k] = P25FE_CLK_INTERP[k];
        phase_sync();
    }
    // detection k: lane k % 64 of the batch in registers (a tile with more than 64 detections reloads, 64 at a time)
    unsigned evr = el.ev, egr = el.eg;
    int batch = 0;
    auto ev_at = [&](int k, unsigned& eg) -> int {
        if ((k & ~(WV - 1)) != batch) {                             // uniform
            batch = k & ~(WV - 1);
            const bool in = batch + lane < EVCAP;
            evr = in ? (unsigned)evl[batch + lane] : 0u;
            egr = in ? evg[batch + lane] : 0u;
        }
        eg = (unsigned)rdl((int)egr, k & (WV - 1));
        return rdl((int)evr, k & (WV - 1));
    };
    uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride + so.dibit_off;
    uint8_t* out2 = a.dibits2 ? a.dibits2 + (size_t)ch * a.dibit_stride + so.dibit_off : nullptr;
    const long room = a.dibit_stride - (long)so.dibit_off;
    const long T0 = a.abs0 + t0, TE = T0 + tn;
    // The tile's samples, by tile-local index d (>= -1 with the interpolator's first tap): planar position t0 + PLPAD + d, and t0 + PLPAD
    // is a whole number of blocks -- counted from ONE BLOCK EARLIER every index below is a small non-negative int (planar_index's
    // arithmetic, without its 64-bit divisions)
    static_assert(TS % PL_BLK == 0 && PLPAD % PL_BLK == 0 && PLPAD >= PL_BLK, "a tile starts on a block of the planar layout");
    const float* fb = f + ((t0 + PLPAD) / PL_BLK - 1) * (long)PL_BLK;
    auto sample_at = [&](int e) -> float {                           // e = d + PL_BLK
        const int sy = e / SPS, r = e - sy * SPS;
        return fb[(sy >> 5) * PL_BLK + r * 32 + (sy & 31)];
    };

    // instants of the clock (s, D, N) with index in [lo_, hi_) that it governs
    auto emit = [&](long s, int D, int N, long lo_, long hi_, float h, float m, float l, int rank) -> int {
        if (lo_ < s + W + 1) lo_ = s + W + 1;
        if (hi_ <= lo_) return 0;
        const long j_lo = clock_J(lo_ - s, D, N) + 1, j_hi = clock_J(hi_ - s, D, N);     // inclusive
        long cnt_l = j_hi - j_lo + 1;                               // a tile holds at most TS / 6 instants under any clock the library makes;
        cnt_l = cnt_l < 0 ? 0 : (cnt_l > (long)TS ? (long)TS : cnt_l);   // clamp whatever a foreign anchor got past the plausibility test
        const int count = (int)cnt_l;
        // instant j sits at s + (j D) div N, phase ((j D) mod N) 64 div N.  With j = j_lo + idx: (j_lo D) div / mod N once per call (uniform),
        // then 32-bit arithmetic per instant whenever (j_lo D) mod N + (count - 1) D fits 32 bits -- it does for every interval between sync
        // words shorter than ~1.4 M samples; longer ones (the spec allows 2^24) and foreign anchors keep the 64-bit form.  Same integers.
        const long B = j_lo * (long)D;
        long qb = B, rr = 0;                                        // N == 1: the nominal clock, no division
        if (N != 1) { qb = B / (long)N; rr = B - qb * (long)N; }
        const long ib = s + qb;                                     // position of instant j_lo
        const unsigned rb = (unsigned)rr;
        const bool narrow = B >= 0 && D > 0 && N < (1 << 26)
                            && (unsigned long long)rr + (unsigned long long)(count > 0 ? count - 1 : 0) * (unsigned long long)D < (1ull << 32);
        uint8_t* dst = out + rank;
        for (int j0 = 0; j0 < count; j0 += WV) {
            const int idx = j0 + lane;
            if (idx < count) {
                long i; int q = 0;
                if (narrow) {                                        // uniform
                    const unsigned t = rb + (unsigned)idx * (unsigned)D;
                    unsigned qu = t;
                    if (N != 1) {
                        qu = t / (unsigned)N;
                        q = (int)(((t - qu * (unsigned)N) * (unsigned)P25FE_CLK_PHASES) / (unsigned)N);
                    }
                    i = ib + (long)qu;
                } else {
                    const long num = (j_lo + idx) * (long)D;
                    const long qu = num / N;
                    i = s + qu; q = (int)(((num - qu * N) * P25FE_CLK_PHASES) / N);
                }
                float v;
                if (i < lo_ || i >= hi_) {
                    // cannot happen for a clock and a position the library produced; a foreign anchor whose position is so far away
                    // that j D wrapped 64 bits (p25fe_slice_dev's d_anchor_in) must not become a load outside the planes
                    v = 0.f;
                } else {
                    const int e = (int)(i - T0) + PL_BLK;            // lo_ >= T0: in [PL_BLK, PL_BLK + TS)
                    if (track) {
                        const float* w = CI + 4 * (q & (P25FE_CLK_PHASES - 1));      // (the mask: a no-op for any clock the library makes)
                        v = w[0] * sample_at(e - 1);
                        v = __builtin_fmaf(w[1], sample_at(e), v);
                        v = __builtin_fmaf(w[2], sample_at(e + 1), v);
                        v = __builtin_fmaf(w[3], sample_at(e + 2), v);
                    } else {
                        v = sample_at(e);
                    }
                }
                if (rank >= 0 && rank + idx < room) {
                    const unsigned char db = slice_dibit(v, h, m, l);
                    dst[idx] = db;
                    if (out2) out2[rank + idx] = db;
                }
            }
        }
        return count;
    };

    int rank = 0;
    if (valid) rank += emit(so.s, so.D, so.N, T0, g.pre_end1 ? T0 + (long)g.pre_end1 - 1 : TE, hi, mid, lo, 0);
    unsigned eg = 0u, eg_prev = 0u;
    int ev_prev = 0;
    for (int k = 0; k < n_ev; ++k) {
        const int evk = ev_at(k, eg);
        const long ek = T0 + evk, sk = ek - W;
        int D, N;
        if (k == 0) clock_period(track, ((g.n_det_flags >> 16) & G_FIRST_TRACKS) && valid, so.s, so.f, sk, frac3(eg >> 16), D, N);
        else clock_period(track, ((eg >> 15) & 1u) != 0, T0 + ev_prev - W, frac3(eg_prev >> 16), sk, frac3(eg >> 16), D, N);
        float h, m, l;
        if (k < EVTHR_N) {
            h = rdl_f(el.th, 3 * k); m = rdl_f(el.th, 3 * k + 1); l = rdl_f(el.th, 3 * k + 2);
        } else {
            float v[NSYN];
            sync_gather(f, t0 + evk - W + PLPAD, v);
            sync_thresholds(v, h, m, l);
        }
        if (lane == 0 && a.sync_pos && (long)(so.event_off + k) < a.sync_stride) {
            a.sync_pos[(size_t)ch * a.sync_stride + so.event_off + k] = sk;
            a.sync_dibit[(size_t)ch * a.sync_stride + so.event_off + k] = so.dibit_off + (unsigned long long)rank;
        }
        rank += emit(sk, D, N, ek + 1, T0 + (long)(eg & 0x7fffu), h, m, l, rank);
        ev_prev = evk; eg_prev = eg;
    }
}
__global__ __launch_bounds__(WV, 4) void k_slice_g(SliceArgsG a) { slice_g_item(a, (int)blockIdx.x, (int)blockIdx.y); }
#endif

// ------------------------------------------------------------------------------------------
// SPEC 3.8c (symbol_clock = 2), resident ranges: the slicer by DETECTION instead of by tile.
//
// A detection without a usable clock of its own (the first of a lock run) takes the clock of the interval that STARTS at it: the
// backward clock of the next detection.  That look-ahead crosses tiles, and the clock changes how many instants the detection
// governs, i.e. every later dibit's offset -- the tile-local bookkeeping of K3 / k_slice_g has no place for it.  The list of
// detections has: K2 and the general receiver's K3 (unchanged) give every tile its carry-in and its first detection's index in the range;
//   k_ev_collect  one wave per tile: its detections -> EvRec[1 + index] (position, backward clock + usable, where the governed
//                 interval ends inside the tile, thresholds); the tile that ends an interval left OPEN by an earlier tile records
//                 where (EvNext of that detection: position + the call's sequence number, so that nothing has to be cleared);
//                 entry 0 is the range's carry-in anchor;
//   k_ev_count    one lane per detection: final clock (one step of look-ahead), its instants in closed form;
//   k_ev_scan     one wave per channel: prefix sum -> offsets, the range's record, the sync lists;
//   k_ev_slice    one wave per 256 dibits: binary search of the offsets, instant position, 4-tap interpolation, thresholds.
// ------------------------------------------------------------------------------------------
struct EvRec {
    long s;                     // sync position
    long g_lo, g_hi;            // governed interval [g_lo, g_hi): g_hi as known inside the detection's tile (valid unless EV_OPEN)
    int Db, Nb;                 // backward clock (SPEC 3.8b)
    int D2, N2;                 // final clock (k_ev_count)
    unsigned flags;             // EV_*
    float hi, mid, lo;
    int pad_;
};
static_assert(sizeof(EvRec) == 64, "EvRec layout");
struct EvNext { long pos; unsigned long long seq; };
constexpr unsigned EV_USABLE = 1u, EV_OPEN = 2u, EV_VALID = 4u;     // bits 8..10: the sync position's fraction (quarter samples, two's complement)
// SPEC 3.8c's fractional anchor: the instants of a detection (s, f) under the clock D / N (N a multiple of 4) are s + (j D + f N / 4) div N.
// number of instants j >= 1 with position < x  (x relative to s): j D + off < x N
__host__ __device__ inline long clock_J_off(long x, int D, int N, int off)
{
    const long num = x * (long)N - (long)off - 1;
    return num < 0 ? 0 : num / (long)D;
}

struct EvArgs {
    Planar pl;
    long n, abs0;
    int n_tiles;
    const ScanOutG* outs;
    const TileSumG* gsum;
    const uint16_t* evl;
    const uint32_t* evg;
    const float* evthr;
    const p25fe_anchor_t* anchor_in;     // nullable, [ch]
    EvRec* rec;                          // [ch][ev_stride]
    EvNext* nxt;                         // [ch][ev_stride]
    unsigned long long* off;             // [ch][ev_stride + 1]
    long ev_stride;
    unsigned long long seq;
    p25fe_result_t* result;              // [ch] (k_scan_tiles_g's; n_sync is read, the counts are rewritten)
    uint8_t* dibits;
    long dibit_stride;
    int64_t* sync_pos;                   // nullable
    uint64_t* sync_dibit;
    long sync_stride;
};

#ifndef P25FE_JIT
__device__ __forceinline__ void ev_collect_item(const EvArgs& a, const int tile, const int ch)
{
    const int lane = threadIdx.x;
    const long t0 = (long)tile * TS;
    const int tn = a.n - t0 < TS ? (int)(a.n - t0) : TS;
    const ScanOutG so = a.outs[(size_t)ch * a.n_tiles + tile];
    const TileSumG g = a.gsum[(size_t)ch * a.n_tiles + tile];
    const int n_ev = (int)(g.n_det_flags & 0xffffu);
    const long T0 = a.abs0 + t0;
    EvRec* rec = a.rec + (size_t)ch * a.ev_stride;
    EvNext* nxt = a.nxt + (size_t)ch * a.ev_stride;
    p25fe_anchor_t Ain;
    Ain.valid = 0; Ain.s = 0; Ain.hi = Ain.mid = Ain.lo = 0.f; Ain.period_d = SPS; Ain.period_n = 1;
    if (a.anchor_in) Ain = a.anchor_in[ch];
    const bool carry_valid = so.src >= 0 || (so.src == -1 && Ain.valid != 0);
    if (tile == 0 && lane == 0) {                                   // entry 0: the range's carry-in anchor (its clock is what it is)
        EvRec r;
        r.s = Ain.s; r.g_lo = a.abs0; r.g_hi = a.abs0 + a.n;
        const bool ok = clock_plausible(Ain.period_d, Ain.period_n);
        r.Db = ok ? Ain.period_d : SPS; r.Nb = ok ? Ain.period_n : 1; r.D2 = r.Db; r.N2 = r.Nb;
        r.flags = EV_USABLE | EV_OPEN | (Ain.valid ? EV_VALID : 0u);
        r.hi = Ain.hi; r.mid = Ain.mid; r.lo = Ain.lo; r.pad_ = 0;
        rec[0] = r;
    }
    if (!g.pre_end1) return;                                        // no event of any kind
    // this tile's first event ends the interval the carry-in's detection left open
    if (lane == 0 && carry_valid) {
        EvNext x; x.pos = T0 + (long)g.pre_end1 - 1; x.seq = a.seq;
        nxt[so.event_off] = x;                                      // (entry of detection event_off - 1; 0 = the carry-in anchor)
    }
    if (n_ev == 0) return;
    const uint16_t* evl = a.evl + ((size_t)ch * a.n_tiles + tile) * EVCAP;
    const uint32_t* evg = a.evg + ((size_t)ch * a.n_tiles + tile) * EVCAP;
    // (one detection per lane: its own entry of K2's lists and its predecessor's, straight from memory -- no LDS)
    const float* f = a.pl.f + (size_t)ch * a.pl.f_ch;
    const float* eth = a.evthr + ((size_t)ch * a.n_tiles + tile) * (EVTHR_N * 3);
    for (int k0 = 0; k0 < n_ev; k0 += WV) {
        const int k = k0 + lane;
        if (k < n_ev) {
            const unsigned eg = evg[k];
            const long ek = T0 + evl[k], sk = ek - W;
            const unsigned eg_p = k ? evg[k - 1] : 0u;
            const long s_p = k ? T0 + evl[k - 1] - W : 0;
            EvRec r;
            bool us;
            if (k == 0) clock_period(true, ((g.n_det_flags >> 16) & G_FIRST_TRACKS) && carry_valid, so.s, so.f, sk, frac3(eg >> 16), r.Db, r.Nb, &us);
            else clock_period(true, ((eg >> 15) & 1u) != 0, s_p, frac3(eg_p >> 16), sk, frac3(eg >> 16), r.Db, r.Nb, &us);
            r.s = sk; r.g_lo = ek + 1; r.g_hi = T0 + (long)(eg & 0x7fffu);
            r.D2 = r.Db; r.N2 = r.Nb;
            r.flags = EV_VALID | (us ? EV_USABLE : 0u) | ((k == n_ev - 1 && (int)(eg & 0x7fffu) == tn) ? EV_OPEN : 0u) | (((eg >> 16) & 7u) << 8);
            r.hi = r.mid = r.lo = 0.f; r.pad_ = 0;
            EvRec* dst = rec + 1 + so.event_off + k;
            if (k < EVTHR_N) {
                r.hi = eth[3 * k]; r.mid = eth[3 * k + 1]; r.lo = eth[3 * k + 2];
                *dst = r;
            } else {
                // (the thresholds of these are written by the loop below: no second store to the same words from this wave)
                dst->s = r.s; dst->g_lo = r.g_lo; dst->g_hi = r.g_hi; dst->Db = r.Db; dst->Nb = r.Nb; dst->D2 = r.D2; dst->N2 = r.N2;
                dst->flags = r.flags; dst->pad_ = 0;
            }
        }
    }
    // thresholds beyond the ones K2 handed over: recomputed from the sync word, one detection at a time (every lane, same window)
    for (int k = EVTHR_N; k < n_ev; ++k) {
        float v[NSYN], h, m, l;
        sync_gather(f, t0 + (int)evl[k] - W + PLPAD, v);
        sync_thresholds(v, h, m, l);
        if (lane == 0) { EvRec* r = rec + 1 + so.event_off + k; r->hi = h; r->mid = m; r->lo = l; }
    }
}
__global__ __launch_bounds__(WV, 4) void k_ev_collect(EvArgs a) { ev_collect_item(a, (int)blockIdx.x, (int)blockIdx.y); }

// k_ev_count: one lane per detection (grid-stride), the divisions of clock_count in parallel; k_ev_scan: ONE wave (it runs beside the
// next call's K1, whose one-wave workgroups leave room for exactly that -- a 512-thread workgroup waits for K1 to drain, as rounds
// 3 - 5's k_scan_g did), prefix sum of the counts, the range's record and the sync lists.
__global__ __launch_bounds__(WV, 4) void k_ev_count(EvArgs a)
{
    const int lane = threadIdx.x, ch = blockIdx.y;
    EvRec* rec = a.rec + (size_t)ch * a.ev_stride;
    const EvNext* nxt = a.nxt + (size_t)ch * a.ev_stride;
    const long n_e = (long)a.result[ch].n_sync + 1;                  // entries 0 .. n_e - 1 (0 = the carry-in anchor)
    const long range_end = a.abs0 + a.n;
    for (long e = (long)blockIdx.x * WV + lane; e < n_e; e += (long)gridDim.x * WV) {
        const EvRec r = rec[e];
        unsigned cnt = 0u;
        long jlo = 1;
        if (r.flags & EV_VALID) {
            int D = r.Db, N = r.Nb;
            if (!(r.flags & EV_USABLE) && e + 1 < n_e) {
                // the next detection's backward clock is the interval that starts here -- if lock was held and it is plausible
                // (fields k_ev_collect wrote; that entry's owner writes D2 / N2 / g_lo / g_hi / pad_ only)
                if (rec[e + 1].flags & EV_USABLE) { D = rec[e + 1].Db; N = rec[e + 1].Nb; }
            }
            long ghi = r.g_hi;
            if (r.flags & EV_OPEN) {
                const EvNext x = nxt[e];
                ghi = x.seq == a.seq ? x.pos : range_end;
            }
            // the fractional anchor (entry 0, the carry-in, arrives without a fraction: bits 8..10 are 0); 10 / 1 is written 40 / 4
            if (N == 1) { D *= 4; N = 4; }
            const int off = frac3(r.flags >> 8) * (N / 4);
            long glo = r.g_lo < r.s + W + 1 ? r.s + W + 1 : r.g_lo;
            const long j0 = clock_J_off(glo - r.s, D, N, off);
            long c = ghi > glo ? clock_J_off(ghi - r.s, D, N, off) - j0 : 0;
            cnt = (unsigned)(c < 0 ? 0 : (c > 0x7fffffffL ? 0x7fffffffL : c));
            jlo = j0 + 1;
            rec[e].D2 = D; rec[e].N2 = N; rec[e].g_hi = ghi;
        }
        rec[e].g_lo = jlo;                                           // from here on: the index j of the first instant this detection governs
        rec[e].pad_ = (int)cnt;
    }
}

constexpr int EVC_PER = 16;
__global__ __launch_bounds__(WV) void k_ev_scan(EvArgs a)
{
    const int lane = threadIdx.x, ch = blockIdx.x;
    const EvRec* rec = a.rec + (size_t)ch * a.ev_stride;
    unsigned long long* off = a.off + (size_t)ch * (a.ev_stride + 1);
    const long n_e = (long)a.result[ch].n_sync + 1;
    unsigned long long c_base = 0ull;                                // uniform
    for (long c0 = 0; c0 < n_e; c0 += (long)WV * EVC_PER) {
        unsigned cnt[EVC_PER];
        unsigned long long mine = 0;
#pragma unroll
        for (int q = 0; q < EVC_PER; ++q) {
            const long e = c0 + (long)lane * EVC_PER + q;
            cnt[q] = e < n_e ? (unsigned)rec[e].pad_ : 0u;
            mine += cnt[q];
        }
        const unsigned long long incl = wave_incl_sum64(mine, lane);
        unsigned long long o = c_base + incl - mine;
#pragma unroll
        for (int q = 0; q < EVC_PER; ++q) {
            const long e = c0 + (long)lane * EVC_PER + q;
            if (e < n_e) {
                off[e] = o;
                if (e >= 1 && a.sync_pos && e - 1 < a.sync_stride) {
                    a.sync_pos[(size_t)ch * a.sync_stride + e - 1] = rec[e].s;
                    a.sync_dibit[(size_t)ch * a.sync_stride + e - 1] = o;
                }
                o += cnt[q];
            }
        }
        c_base += __shfl(incl, WV - 1, 64);
    }
    if (lane == 0) {
        off[n_e] = c_base;
        p25fe_result_t r = a.result[ch];
        r.n_dibits = c_base;
        r.n_dibits_after_first = n_e >= 2 ? c_base - off[1] : 0;
        a.result[ch] = r;
    }
}

__global__ __launch_bounds__(WV, 4) void k_ev_slice(EvArgs a)
{
    __shared__ float CI[P25FE_CLK_PHASES * 4];
    const int lane = threadIdx.x, ch = blockIdx.y;
    const EvRec* rec = a.rec + (size_t)ch * a.ev_stride;
    const unsigned long long* off = a.off + (size_t)ch * (a.ev_stride + 1);
    const long n_e = (long)a.result[ch].n_sync + 1;
    const unsigned long long total = off[n_e];
    const unsigned long long g0 = (unsigned long long)blockIdx.x * (WV * 4);
    if (g0 >= total) return;
    for (int k = lane; k < P25FE_CLK_PHASES * 4; k += WV) CI[k] = P25FE_CLK_INTERP[k];
    phase_sync();
    const float* f = a.pl.f + (size_t)ch * a.pl.f_ch;
    uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride;
    // the detection of the wave's first dibit: one binary search with uniform indices (scalar loads); lanes then walk forward
    long e0 = 0;
    {
        long hi_ = n_e;                                             // largest e with off[e] <= g0
        while (hi_ - e0 > 1) {
            const long mid_ = (e0 + hi_) >> 1;
            if (off[mid_] <= g0) e0 = mid_; else hi_ = mid_;
        }
    }
    long lo_ = e0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long gi = g0 + (unsigned long long)(q * WV + lane);
        if (gi < total && (long)gi < a.dibit_stride) {
            while (lo_ + 1 < n_e && off[lo_ + 1] <= gi) ++lo_;      // (864 dibits per detection on a P25 channel: zero or one step)
            const EvRec r = rec[lo_];
            const long j = r.g_lo + (long)(gi - off[lo_]);           // (k_ev_count left the first governed instant's index in g_lo)
            const long num = j * (long)r.D2 + (long)(frac3(r.flags >> 8) * (r.N2 / 4));    // (> 0: j >= 1 and |offset| <= N / 2 < D)
            const long qu = num / r.N2;
            const long i = r.s + qu;
            // (the remainder is < N2 <= 2^26 -- clock_plausible -- so the phase is a 32-bit division)
            const int ph = (int)((((unsigned)(num - qu * r.N2) * (unsigned)P25FE_CLK_PHASES) / (unsigned)r.N2) & (unsigned)(P25FE_CLK_PHASES - 1));   // (the mask: a no-op for any clock the library makes)
            // an instant lies inside the range by construction; a foreign carry-in anchor whose position is so far away that j D
            // wrapped (anchors also arrive from outside: p25fe_slice_dev's d_anchor_in) must not turn into a load outside the planes
            if (i < a.abs0 || i >= a.abs0 + a.n) { out[gi] = 0; continue; }
            const long p = i - a.abs0 + PLPAD;
            float b[4];
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const long pp = p - 1 + tq;
                const long sy = pp / SPS;
                b[tq] = f[planar_index(sy, (int)(pp - sy * SPS))];
            }
            const float* w = CI + 4 * ph;
            float v = w[0] * b[0];
            v = __builtin_fmaf(w[1], b[1], v);
            v = __builtin_fmaf(w[2], b[2], v);
            v = __builtin_fmaf(w[3], b[3], v);
            out[gi] = slice_dibit(v, r.hi, r.mid, r.lo);
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Streaming chunks (the reference's unit of work: one 32 768-byte read of the dongle = 3 276 / 3 277 baseband samples,
// src/consts.rs:6-8, src/demod.rs:87-90): a range of at most ONE tile runs the whole receiver -- sync detection, the
// (trivial) scan, the slicer -- in one wave, with the carry-in anchor read from and every result written to host-visible
// memory, so that a chunk costs one kernel launch and one synchronisation instead of five launches and six copies.
// Fixed-stride clock only (the tracking clock and in-range lock drops take the general kernels).
// ------------------------------------------------------------------------------------------
constexpr int TAILN = 256;                                          // newest baseband samples handed back to the host (>= HIST_BB + CLK_L)

struct ChunkRecvArgs {
    Planar pl;                  // planar scratch holding the range (written by K1 or by planarize_one below)
    long n;                     // processed baseband samples per channel, <= TS
    long abs0;                  // absolute index of the first one
    TileRec* recs;              // [ch] one tile of K2's scratch per channel
    unsigned long long* tsum;   // [ch]
    uint16_t* evl;              // [ch][EVCAP]
    float* evthr;               // [ch][EVTHR_N][3]
    const p25fe_anchor_t* anchor_in;    // [ch]
    p25fe_result_t* result;     // [ch]
    uint8_t* dibits;            // [ch][dibit_stride]
    long dibit_stride;
    int64_t* sync_pos;          // nullable
    uint64_t* sync_dibit;
    long sync_stride;
    float* tail;                // nullable: [ch][TAILN] newest baseband samples of the range, oldest first (entries older than the
                                // recomputed history are left untouched)
    int look;                   // baseband sample m of the range sits at planar position m + look + PLPAD
    unsigned long long n_baseband;
    unsigned* done;             // nullable: [ch] in host-visible memory, receives `seq` after everything else of the channel has
    unsigned seq;               // been written (system-scope release): the host polls it instead of synchronising the stream
};

// every lane's global stores of the wave are visible to every lane's later loads (one wave: same CU, same L1)
__device__ __forceinline__ void wave_global_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// the newest TAILN baseband samples of the range, read back from the planes (what a later p25fe_slice call on the handle
// needs as its history): owned sample m sits at planar position m + look + PLPAD, K1 wrote m >= -(HIST_BB + look)
__device__ __forceinline__ void tail_extract(const ChunkRecvArgs& c, const int ch)
{
    if (!c.tail) return;
    const float* f = c.pl.f + (size_t)ch * c.pl.f_ch;
    for (int j = (int)threadIdx.x; j < TAILN; j += WV) {
        const long m = c.n - TAILN + j;
        if (m >= -(long)(HIST_BB + c.look)) {
            const long pp = m + c.look + PLPAD;
            const long sy = pp / SPS;
            c.tail[(size_t)ch * TAILN + j] = f[planar_index(sy, (int)(pp - sy * SPS))];
        }
    }
}
#ifndef P25FE_JIT
__global__ __launch_bounds__(WV) void k_tail_extract(ChunkRecvArgs c) { tail_extract(c, (int)blockIdx.x); }
#endif

__device__ __forceinline__ void recv_one_tile(const ChunkRecvArgs& c, const int ch)
{
    const int lane = threadIdx.x;
    DetArgs d;
    d.pl = c.pl; d.n = c.n; d.abs0 = c.abs0; d.n_tiles = 1;
    d.recs = c.recs; d.tsum = c.tsum; d.evl = c.evl; d.evthr = c.evthr;
    d.opt.track = 0; d.opt.n_resync = 0; d.opt.resync = nullptr; d.opt.resync_stride = 0; d.gsum = nullptr; d.evg = nullptr;
    d.head_flag = nullptr; d.head_seq = 0u; d.head_tile_max = -1; d.head_err = nullptr;
    detect_tile<false>(d, 0, ch);
    wave_global_sync();
    // the scan of a one-tile range
    const TileRec rc = c.recs[ch];
    const unsigned long long u = c.tsum[ch];
    p25fe_anchor_t A = c.anchor_in[ch];
    if (!clock_plausible(A.period_d, A.period_n)) { A.period_d = SPS; A.period_n = 1; }
    const bool own = rc.n_events > 0;
    const unsigned long long pre = A.valid ? (unsigned long long)count_instants(A.s, c.abs0, own ? rc.first_event + 1 : c.abs0 + c.n) : 0ull;
    if (lane == 0) {
        p25fe_result_t r;
        r.n_baseband = c.n_baseband;
        r.n_dibits = pre + (own ? (unsigned long long)rc.post_count : 0ull);
        r.n_sync = own ? (unsigned long long)rc.n_events : 0ull;
        p25fe_anchor_t out = A;
        if (own) { out.valid = 1; out.s = rc.last_s; out.hi = rc.hi; out.mid = rc.mid; out.lo = rc.lo; out.period_d = SPS; out.period_n = 1; }
        r.anchor_out = out;
        r.first_event = own ? rc.first_event : -1;
        r.n_dibits_after_first = own ? (unsigned long long)rc.post_count : 0ull;
        r.carry_end = own ? rc.first_event + 1 : -1;
        r.first_seg_end = -1; r.flags = 0u; r.reserved = 0u;
        c.result[ch] = r;
    }
    SliceArgs l;
    l.pl = c.pl; l.n = c.n; l.abs0 = c.abs0; l.n_tiles = 1;
    l.outs = nullptr; l.recs = c.recs; l.tsum = c.tsum; l.evl = c.evl; l.evthr = c.evthr; l.anchor_in = c.anchor_in;
    l.dibits = c.dibits; l.dibit_stride = c.dibit_stride;
    l.sync_pos = (c.sync_pos && c.sync_dibit) ? c.sync_pos : nullptr; l.sync_dibit = c.sync_dibit; l.sync_stride = c.sync_stride;
    l.dibits2 = nullptr; l.fix.summ = nullptr; l.gpre = nullptr;
    ScanOut so;
    so.src = -1; so.event_off = 0u; so.dibit_off = 0ull;
    slice_tile(l, 0, ch, so, u, A.valid, A.s, A.hi, A.mid, A.lo, ev_lanes_load(c.evl + (size_t)ch * EVCAP, nullptr, c.evthr + (size_t)ch * (EVTHR_N * 3)));
    tail_extract(c, ch);
    if (c.done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");              // system scope: the host sees dibits, result and tail before the flag
        if (lane == 0) __hip_atomic_store(&c.done[ch], c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// p25fe_slice for a chunk of at most one tile: planarize (one wave walks the ten planes), then the receiver.
struct RecvChunkArgs {
    ChunkRecvArgs r;
    const float* bb;            // owned baseband sample 0 of channel 0 (host-visible staging: [tail | new])
    long bb_stride;
    long n_hist;                // valid samples before it
    float* f;                   // planar scratch, writable view of r.pl
    uint32_t* bits;
    long n_blocks;
};

#ifndef P25FE_JIT
__global__ __launch_bounds__(WV, 4) void k_recv_chunk(RecvChunkArgs a)
{
    const int lane = threadIdx.x, ch = blockIdx.x;
    const long hist = a.n_hist < HIST_BB + a.r.look ? a.n_hist : HIST_BB + a.r.look;
    const float* bb = a.bb + (size_t)ch * a.bb_stride;
    float* f = a.f + (size_t)ch * a.r.pl.f_ch;
    uint32_t* bits = a.bits + (size_t)ch * a.r.pl.bits_ch;
    // symbols that hold data (whole 64-symbol strides; planes past the range are never decided on: K2 drops positions >= tn)
    long n_sym = ((a.r.n + a.r.look + PLPAD + SPS - 1) / SPS + WV - 1) / WV * WV;
    if (n_sym > a.n_blocks * 32) n_sym = a.n_blocks * 32;
    for (long i0 = 0; i0 < n_sym; i0 += WV) {
        const long i = i0 + lane;
#pragma unroll
        for (int r = 0; r < SPS; ++r) {
            const long m = SPS * i + r - PLPAD - a.r.look;
            const float v = (m >= -hist && m < a.r.n + a.r.look) ? bb[m] : 0.0f;
            f[planar_index(i, r)] = v;
            const unsigned long long sg = __ballot(__float_as_int(v) < 0);
            if ((lane & 31) == 0) bits[(i >> 5) * SPS + r] = (unsigned)(sg >> (lane & 32));
        }
    }
    wave_global_sync();
    recv_one_tile(a.r, ch);
}
#endif

// ------------------------------------------------------------------------------------------
// linear baseband -> planes + sign bits (entry points that receive a 48 kHz float stream).  Workgroup = 10 waves,
// wave r fills 64 symbols of plane r: the ten waves read the same 2.5 KB of the linear stream (L1), every plane row
// is written as one contiguous 256-byte store, and the ballot of the sign bits is the plane's 64-bit word.
// ------------------------------------------------------------------------------------------
struct PlanarizeArgs {
    const float* bb;        // owned baseband sample 0 of channel 0
    long bb_stride;
    long n_hist;            // valid samples before it
    long n;                 // owned samples
    float* f;               // planar out, channel 0
    long f_ch;
    uint32_t* bits;
    long bits_ch;
    long n_blocks;          // blocks per channel
    int shift;              // sample m goes to planar position m + shift + PLPAD (the general receiver's lookahead)
};

#ifndef P25FE_JIT
__global__ __launch_bounds__(WV * SPS) void k_planarize(PlanarizeArgs a)
{
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6, ch = blockIdx.y;
    const long i = (long)blockIdx.x * WV + lane;                    // two blocks per workgroup
    const long m = SPS * i + r - PLPAD - a.shift;
    const long hist = a.n_hist < HIST_BB + a.shift ? a.n_hist : HIST_BB + a.shift;
    const float v = (m >= -hist && m < a.n) ? a.bb[(size_t)ch * a.bb_stride + m] : 0.0f;
    const bool in = (i >> 5) < a.n_blocks;
    if (in) a.f[(size_t)ch * a.f_ch + planar_index(i, r)] = v;
    const unsigned long long sg = __ballot(__float_as_int(v) < 0);
    if ((lane & 31) == 0 && in) a.bits[(size_t)ch * a.bits_ch + (i >> 5) * SPS + r] = (unsigned)(sg >> (lane & 32));
}
#endif

}  // namespace p25k
