// p25fe_kernels.hip -- CDNA4 (gfx950) kernels of the P25 front end.
//
// Hot path of kchmck/p25rx re-designed for MI355X (stages of SURVEY.md section 8a):
//   K1 k_frontend : [u8 -> cf32] -> 5:1 decimating FIR -> channel FIR -> FM discriminator ->
//                   10-sample boxcar  == the five per-chunk loops of DemodTask::run
//                   (src/demod.rs:82-84, 87, 93, 109-111, 114) fused into one pass over HBM.  Writes the baseband
//                   either linearly (the RecvEvent::Baseband hand-off) or in the blocked polyphase layout with one sign
//                   bit per sample on the side (the fused path: PLPAD / planar_index below).
//   K2 k_detect, K3 k_scan_tiles, K4 k_slice (p25fe_recv.hip, included below): frame-sync detection, the receiver's serial
//                   state as a scan, 4-level slicer -- the front half of MessageReceiver::feed (src/recv.rs:207).
// Either side of the path (SURVEY.md section 8f / BASELINE.json config 3):
//   K0 k_predecim   : 2.4 Msps -> 240 ksps, 80-tap 10:1 decimating FIR (config 3's extra stage).
//   K5 k_nid        : network identifier after each frame sync, exhaustive BCH(63,16,23) search; k_chan_stats folds the
//                     records into the reference's per-channel statistics shape (src/hub.rs:557-581).
//   K6 k_channelise : one 2.4 Msps capture -> 192 channel streams at 240 ksps (factored 192-point DFT in registers).
//   k_shard_resolve / k_shard_compact : carry resolution and dibit compaction across time shards (config 5).
// K0, K1, K2, K4, K6 (everything that touches sample data): one wave per workgroup, no s_barrier, short-lived workgroups --
// measured fastest for each of them.  K3 and K5 are small block-wide scans / searches.
//
// Arithmetic contract (docs/SPEC.md section 3): fp32, fma only where the spec says fma, single
// accumulator per output in tap order 0..T-1.  Compiled with -ffp-contract=off.
// Stencil / element-wise work: no MFMA.  HBM-bound by design: 8 B in + 0.8 B out per IQ sample.
// The same file is the source of the SPECIALISED kernels: p25fe_jit.cpp hands it to hipRTC (which brings its own runtime
// declarations and no system headers) behind a generated header that defines P25FE_JIT and the caller's numbers.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "p25fe.h"
#include "p25fe_spec.h"

namespace p25k {

// ---- measurement hooks -------------------------------------------------------------------------------------------------------------
// The product's kernels -- and the source hipRTC compiles for a caller's numbers -- hold NO measurement code.  The builds that time
// parts of K1 / K2 / K6 (tools/ablate.sh, tools/k1_stamps.py, tools/k6_variants.sh, the model of docs/K1_MODEL.md: `make variant
// DEFS="-DP25FE_MEASURE ..."`) get the bodies of the hooks below from p25fe_measure.inc, which is neither part of the library's
// embedded source nor read by the product build.  Here every hook is empty (or the constant `false`), so the product's ISA does not
// depend on them (tests/test_isa_lint.py; the kernels' text is identical to that of the build that had the code inline).
#ifdef P25FE_MEASURE
#include "p25fe_measure.inc"
#else
#define P25FE_M_LOAD_OFFSET(product_, cached_row_) (product_)      // the window loader's byte offset
#define P25FE_M_ENTRY do { } while (0)                              // frontend_body: entry
#define P25FE_M_LOCALS do { } while (0)                             // ... its counters
#define P25FE_M_ITEM do { } while (0)                               // ... top of a work item
#define P25FE_M_SUBTILE do { } while (0)                            // ... top of a sub-tile
#define P25FE_M_EXIT do { } while (0)                               // ... exit
#define K1_STAMP(i) do { } while (0)                                // phase boundary i of a sub-tile
#define K1_PIN2(a, b) do { } while (0)
#define P25FE_M_PIN_V2(arr_, n_) do { } while (0)
#define P25FE_M_PIN_F2(arr_, n_) do { } while (0)
#define P25FE_M_CUT_LOADS do { } while (0)                          // truncated builds: leave the sub-tile after stage N
#define P25FE_M_CUT(stage_) do { } while (0)
#define P25FE_M_CUT_KEEP_F2(stage_, arr_, n_) do { } while (0)
#define P25FE_M_CUT_KEEP_F(stage_, arr_, n_) do { } while (0)
#define P25FE_M_KEEP1(a_) do { } while (0)
#define P25FE_M_KEEP2(a_, b_) do { } while (0)
#define P25FE_M_K6_STORES_ONLY do { } while (0)
#define P25FE_M_NO_BB_STORES false                                  // parts of the output stream left out
#define P25FE_M_NO_SIGN_PLANES false
#define P25FE_M_NO_CH_FMA false
#define P25FE_M_NO_OUT_STAGE false
#define P25FE_M_DET_CUT_SCREEN do { } while (0)                     // K2 truncated after its screen / without its peak test
#define P25FE_M_DET_NO_PEAK_TEST false
#define P25FE_M_PIPE_NOWAIT() false                                 // host side (p25fe_api.hip)
#define P25FE_M_API_EXTRAS
#endif


// (own two-line versions: hipRTC has no <type_traits>)
template <int V> struct icst { static constexpr int value = V; };
template <bool B, class A, class C> struct cond { using type = A; };
template <class A, class C> struct cond<false, A, C> { using type = C; };

// Numbers of the immediate-coefficient ("CT") kernels: the build's own tables (p25fe_spec.h), or -- in a hipRTC build --
// the caller's, from the generated header (p25fe_jit.cpp: tables already padded to the evaluation length).
#ifdef P25FE_JIT
#define K1_CT_DECIM_TAPS P25FE_JIT_DECIM_TAPS
#define K1_CT_CHAN_TAPS P25FE_JIT_CHAN_TAPS
#define K1_CT_FM_GAIN P25FE_JIT_FM_GAIN
#define K1_CT_U8_SCALE P25FE_JIT_U8_SCALE
#define K1_CT_U8_OFFSET P25FE_JIT_U8_OFFSET
#define K1_CT_U8_LUT P25FE_JIT_U8_LUT             /* 1: the u8 table is not affine -> looked up in LDS */
#define K1_CT_AVG_N P25FE_JIT_AVG_N               /* post-discriminator filter (docs/SPEC.md 3.5): taps, */
#define K1_CT_AVG_TAPS P25FE_JIT_AVG_TAPS
#define K1_CT_AVG_UNIFORM P25FE_JIT_AVG_UNIFORM   /* 1: all taps equal = MovingAverage::new(n): sum, then ONE multiply */
#else
#define K1_CT_DECIM_TAPS P25FE_DEFAULT_DECIM_TAPS
#define K1_CT_CHAN_TAPS P25FE_DEFAULT_CHAN_TAPS
#define K1_CT_FM_GAIN P25FE_FM_GAIN
#define K1_CT_U8_SCALE P25FE_U8_SCALE
#define K1_CT_U8_OFFSET P25FE_U8_OFFSET
#define K1_CT_U8_LUT 0
#define K1_CT_AVG_N P25FE_BOXCAR
#define K1_CT_AVG_TAPS P25FE_DEFAULT_AVG_TAPS
#define K1_CT_AVG_UNIFORM 1
#endif

// ------------------------------------------------------------------------------------------
// geometry of K1.  A workgroup is ONE wave (64 lanes): no s_barrier anywhere, and the CU's resident
// waves (2-3 per SIMD) run their load / FIR / FM phases out of step with each other, which is what
// overlaps HBM latency, LDS traffic and VALU work.
// ------------------------------------------------------------------------------------------
constexpr int WV = 64;                       // lanes per workgroup
constexpr int DEC = P25FE_DECIM;
constexpr int T1 = P25FE_T1;
constexpr int T2 = P25FE_T2;
constexpr int BOX = P25FE_BOXCAR;            // the build's own post-discriminator filter: MovingAverage::new(10), src/demod.rs:52
constexpr int HALO_Y = BOX;                  // y needed from m0-10 (fm needs y[m-1], boxcar needs fm[m-9]) -- with the build's own numbers;
                                             // a kernel's own value is its T3 (frontend_body): the post-discriminator filter's length
constexpr int HALO_D = HALO_Y + (T2 - 1);    // 50: d needed from m0-50 (the build's own tap counts; Geo<PK, 1> has its own)
constexpr int TMAX = P25FE_MAX_TAPS;         // 64: tap-count ceiling of the ABI (p25fe_config_t), the generic kernels' geometry
constexpr int AVG_DPP_MAX = 11;              // post-discriminator filters up to this length run in registers (DPP + SGPR carries: two
                                             // lanes back); longer ones through an LDS window like the other FIRs
// A segment (the consecutive sub-tiles one workgroup walks) needs, in front of its first output, the HALO_Y + (T2 - 1)
// decimator outputs its first channel-filter / discriminator / boxcar results depend on.  Two forms, chosen per input
// format by measurement (seg_prologue() below):
//  * halo (rounds 1-2; cf32): the first sub-tile recomputes SEG_HALO = 80 whole OUTPUTS in front of the segment and drops
//    them -- 80 = one byte of every sign-bit plane (8 symbols x 10 samples), so that with segment lengths that are
//    multiples of 80 no two workgroups share a byte of the polyphase layout below;
//  * prologue (round 3; u8): just those d's, the 10 channel outputs and 9 discriminator values behind them are computed in
//    a short prologue, every sub-tile yields 320 outputs and -- with ranges that start on a block boundary of the layout --
//    IS one 1280-byte block: whole 128-byte rows, one 32-bit sign word per plane.  4 % fewer instructions and input
//    bytes per output.  Same box, same capture (tools/bench_ab.sh, tools/k1_ab.sh): the instruction-bound u8 kernel gains
//    3 % (0.2021 - 0.2029 against 0.2075 - 0.2101 ms); the cf32 kernel, whose time follows neither its instruction count
//    nor its bytes, LOSES 1.5 - 2 % (0.2570 - 0.2590 against 0.2519 - 0.2550 ms: the prologue's
//    registers push it from 156 to 168 - 174 VGPRs) and keeps the halo form.
constexpr int SEG_HALO = 80;                 // ... with post-discriminator filters of up to 17 taps (80 >= T3 + 64 - 1); 160 beyond:
// the halo a segment recomputes, for a post-discriminator filter of t3 taps behind a channel filter evaluated at t2 taps
__host__ __device__ constexpr int seg_halo_for(int t3, int t2) { return t3 + t2 - 1 <= 80 ? 80 : 160; }
static_assert(SEG_HALO == seg_halo_for(BOX, TMAX) && seg_halo_for(TMAX, TMAX) >= TMAX + TMAX - 1, "segment halo covers the filter memory and is byte-aligned per plane");
#ifndef P25FE_K1_PRO_CF32
#define P25FE_K1_PRO_CF32 0
#endif
__host__ __device__ constexpr bool seg_prologue(int fmt) { return fmt == P25FE_FMT_U8 || P25FE_K1_PRO_CF32 != 0; }

// Polyphase ("planar") baseband layout of the fused path: with p = m + PLPAD (m = range-local baseband index, the
// 240 history samples of the receiver at m = -240..-1), sample p belongs to plane r = p % 10 at symbol index i = p / 10.
// Storage is BLOCKED: a block holds 32 consecutive symbols of all ten planes,
//   bbp[(i / 32) * 320 + r * 32 + i % 32]          fp32  (one block = 1280 contiguous bytes = one K1 sub-tile)
//   bits[(i / 32) * 10 + r]  bit (i % 32)           sign of that sample (1 = negative or -0)
// A locked 4800 Bd slicer reads one plane: 128 contiguous bytes per 32 symbols; a 24-symbol sync window is 96 bytes
// in at most two pieces; the sign words (1/32 of the baseband) are all the frame-sync screen has to touch.
constexpr int SPS_ = P25FE_SPS;
constexpr int PLPAD = 320;                   // one block of history in front of owned sample 0 (= bit 0 of word 1); K1 starts at position 0
static_assert(PLPAD % 320 == 0 && PLPAD >= 240 + 2, "planar pad holds the receiver's history");
constexpr int OUT_LINEAR = 0, OUT_PLANAR = 1;
constexpr int PL_BLK = 32 * P25FE_SPS;       // floats per block
__host__ __device__ inline long planar_index(long i, int r) { return (i >> 5) * PL_BLK + r * 32 + (i & 31); }
// history (input samples before the first owned one) needed for exact results
constexpr int HIST_IQ = DEC * HALO_D + (T1 - 1) + (DEC - 1);   // 284
constexpr int HIST_IQ_MAX = DEC * (TMAX + TMAX - 1) + (TMAX - 1) + (DEC - 1);   // 702 with 64 + 64 + 64 taps

// PK = consecutive FIR outputs per lane (odd: lane stride 2*5*PK / 2*PK dwords -> conflict-free ds_read_b64)
// TX = 0: the tap counts of docs/SPEC.md (31 / 41); TX = 1: the ABI's ceiling, 64 / 64, for caller-supplied tables such as
// the reference's own p25_filts::DecimFir / BandpassFir (src/demod.rs:27-29, sizes not in the reference tree).
template <int PK, int TX = 0> struct Geo {
    static constexpr int T1 = TX ? TMAX : P25FE_T1;
    static constexpr int T2 = TX ? TMAX : P25FE_T2;
    static constexpr int D_CARRY = (T2 - 1 + 1) & ~1;     // d carried between sub-tiles (T2 - 1, rounded up to keep the window 16-B aligned)
    static constexpr int P = PK;
    static constexpr int SUB = WV * PK;                   // decimated samples per sub-tile (320 for PK = 5)
    static constexpr int XWIN = DEC * SUB + (T1 - DEC);   // input samples feeding one sub-tile of d
    static constexpr int XIN_N = (XWIN + 1 + 7) / 8 * 8;       // window + the 0 / 1 sample alignment shift, rounded up
    static constexpr int D_N = D_CARRY + SUB;
    // LDS: [d carry 40][window region XIN_N][taps].  The sub-tile's new d samples and the output transpose OVERWRITE the
    // front of the window region once the decimator has consumed it (one wave: program order), so that 11 one-wave
    // workgroups fit a CU's 160 KB instead of 9 (occupancy is what bounds the overlap of HBM, LDS and VALU work).
    // then, as far as the kernel variant needs them (k1_lds_bytes): [taps: decimator | channel | 3 | post-discriminator TMAX + 1]
    // (generic kernels) [u8 table 256] [fm window T3 - 1 + SUB] (post-discriminator filters longer than AVG_DPP_MAX)
    static constexpr int TAPS_N = T1 + T2 + 3 + TMAX + 1;  // floats of the generic kernels' taps area
    static constexpr size_t LDS_BASE = sizeof(float2) * (D_CARRY + XIN_N);
    static_assert(sizeof(float2) * XIN_N >= sizeof(float2) * SUB + sizeof(float) * SUB, "d and the transpose fit the window region");
    static constexpr int WAVES_PER_SIMD = PK <= 3 ? 4 : 2;   // register budget the kernel is compiled for (128 / 256 VGPRs)
};

// What the generic kernels read their numbers from (device memory; the immediate-coefficient kernels read only `lut`, and
// only when the handle's u8 table is not affine)
struct Taps {
    float dec[TMAX];
    float ch[TMAX];
    float lut[256];         // rtlsdr_iq::IQ as one byte -> float table (src/demod.rs:83); I and Q alike
    float fm_gain;          // FmDemod's output scale (src/demod.rs:54)
    int n_avg;              // post-discriminator filter (MovingAverage::new(10), src/demod.rs:52; docs/SPEC.md 3.5): taps in use,
    int avg_uniform;        // 1: all of them equal -> sum newest first, then one multiply by avg[0]; 0: fma chain in tap order
    float pad_;
    float avg[TMAX];
};
#ifndef P25FE_K1_AVG_LDS_CF32
#define P25FE_K1_AVG_LDS_CF32 0            /* measurement builds: the LDS window also for short filters in the cf32 kernels (frees 12 SGPR carries) */
#endif
// dynamic LDS of a K1 workgroup (host and kernel agree through this one formula)
template <class G> __host__ __device__ constexpr size_t k1_lds_bytes(bool ct, bool lut, int t3)
{
    return G::LDS_BASE + sizeof(float) * (size_t)((ct ? 0 : G::TAPS_N) + (lut ? 256 : 0) + (t3 > AVG_DPP_MAX || !ct || P25FE_K1_AVG_LDS_CF32 ? t3 - 1 + G::SUB : 0));
}

// SPEC 3.4: polynomial atan2, identical operation sequence to the oracle's restatement.
__device__ __forceinline__ float spec_atan2f(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    // Branch-free: the spec's "if mx == 0 return +0" is a select at the END (0 / 0 = NaN flows through the polynomial and
    // is discarded).  With the early return every call sat in its own exec-masked block, and a lane's five calls -- a
    // correctly rounded division and a 7-deep Horner chain each -- ran one after the other with no instruction-level
    // parallelism; as straight-line code the scheduler interleaves the five chains.
    const float t = mn / mx;
    const float s = t * t;
    float p = P25FE_ATAN_COEFFS[P25FE_ATAN_NCOEF - 1];
#pragma unroll
    for (int i = P25FE_ATAN_NCOEF - 2; i >= 0; --i) p = __builtin_fmaf(p, s, P25FE_ATAN_COEFFS[i]);
    float r = p * t;
    if (ay > ax) r = P25FE_HALF_PI - r;
    if (x < 0.0f) r = P25FE_PI - r;
    if (y < 0.0f) r = -r;
    return mx == 0.0f ? 0.0f : r;
}

// SPEC 3.4: FM discriminator, angle of s * conj(prev) times fs / (2 pi dev)   (src/demod.rs:54, 110)
__device__ __forceinline__ float fm_discriminate(float2 s, float2 prev, float gain)
{
    const float t = s.y * prev.y;
    const float re = __builtin_fmaf(s.x, prev.x, t);
    const float u = s.x * prev.y;
    const float im = __builtin_fmaf(s.y, prev.x, -u);
    return spec_atan2f(im, re) * gain;
}

// Complex sample as a native 2-vector: fma on it is ONE v_pk_fma_f32 (re and im lanes, each a fused multiply-add, SPEC
// section 2).  Written explicitly so that the packing does not depend on the SLP vectoriser: one build of this kernel
// paired the imaginary parts of two different outputs instead and paid 175 v_mov_b32 per sub-tile to assemble the pairs.
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef P25FE_K1_SCALAR_FMA
#define P25FE_K1_SCALAR_FMA 0
#endif
__device__ __forceinline__ v2f cfma(float tap, v2f s, v2f acc)
{
#if P25FE_K1_SCALAR_FMA
    // two v_fma_f32 (2 cycles each on the 32-wide SIMD) instead of one v_pk_fma_f32; the empty asm keeps the SLP vectoriser
    // from re-packing the pair
    float re = __builtin_fmaf(tap, s.x, acc.x), im = __builtin_fmaf(tap, s.y, acc.y);
    asm volatile("" : "+v"(re));
    return v2f{re, im};
#else
    const v2f t = {tap, tap};
    return __builtin_elementwise_fma(t, s, acc);
#endif
}

// One complex sample from LDS as a single ds_read_b64.  The volatile 64-bit access keeps the compiler from
// pairing neighbouring reads into ds_read2_b64, which moves 16 B per lane at HALF the LDS rate of two
// ds_read_b64 (MI355X_MICROARCH.md LDS table: 8 vs 2+2 cycles per wave-instruction).
__device__ __forceinline__ v2f lds_read_v2(const float2* p)
{
    typedef const volatile unsigned long long __attribute__((address_space(3))) * lds_u64_ptr;
    const unsigned long long u = *((lds_u64_ptr)p);                 // generic -> LDS address space: stays a DS op
    return __builtin_bit_cast(v2f, u);
}
__device__ __forceinline__ float2 lds_read_c(const float2* p)
{
    const v2f v = lds_read_v2(p);
    return make_float2(v.x, v.y);
}

// Software-pipelined window reads.  The FIR inner loops are chains "ds_read_b64 -> up to P packed FMAs"; scheduled by the
// compiler (volatile reads above) only one or two reads are in flight, so every step pays most of the LDS latency
// (64 cycles idle, several hundred with eleven waves hammering the same LDS) and the 2-3 waves of a SIMD cannot cover
// 96 such stalls per sub-tile.  Here the reads are issued from inline asm -- which the compiler's s_waitcnt insertion does
// not track -- DEPTH reads ahead of their use, and each use is preceded by an explicit s_waitcnt lgkmcnt(reads issued
// after it).  LDS operations of a wave return in order, so "at most n outstanding" means everything older than the
// last n has landed; operations the compiler interleaves on its own only make the wait stricter.
// The wait takes the value as an in/out operand: its consumers depend on the wait, not on the issue.
template <int OFF> __device__ __forceinline__ v2f lds_issue_b64(unsigned addr)
{
    v2f v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void lds_landed(v2f& v)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(icst<I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// Walk w[N-1], w[N-2], ..., w[0] (newest sample first = tap order 0..T-1), DEPTH reads in flight: body(j, value of w[j]).
template <int N, int DEPTH, class Body> __device__ __forceinline__ void lds_walk_down(const float2* w, Body&& body)
{
    static_assert(DEPTH >= 1 && DEPTH <= 16 && DEPTH <= N, "ring depth");
    const unsigned base = lds_addr(w);
    v2f ring[DEPTH];
    static_for<0, DEPTH>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ring[i] = lds_issue_b64<8 * (N - 1 - i)>(base);
    });
    static_for<0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int after = (N - 1 - i) < (DEPTH - 1) ? (N - 1 - i) : (DEPTH - 1);      // reads issued after read i by now
        lds_landed<after>(ring[i % DEPTH]);
        body(icst<N - 1 - i>{}, ring[i % DEPTH]);
        if constexpr (i + DEPTH < N) ring[i % DEPTH] = lds_issue_b64<8 * (N - 1 - (i + DEPTH))>(base);
    });
}

// The same walk with the reads in groups of G and ONE s_waitcnt per group (two groups of registers: group g + 1 is
// requested before group g is waited for).  A wave issues about one instruction of any kind per 5 cycles (a lone wave
// per CU needs 6 600 cycles for the ~1 300 instructions of a sub-tile), and the per-read s_waitcnt -- plus the s_nop
// the compiler puts behind every inline-asm statement that defines a VALU operand -- were 190 of them.
template <int G, int AFTER> __device__ __forceinline__ void lds_landed_group(v2f (&r)[G])
{
    static_assert(G == 8 || G == 6 || G == 4, "group size");
    static_assert(AFTER >= 0 && AFTER <= 15, "lgkmcnt is a 4-bit field");
    if constexpr (G == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(AFTER));
    else if constexpr (G == 6)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(AFTER));
    else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(AFTER));
}
template <int N, int G, class Body> __device__ __forceinline__ void lds_walk_down_grouped(const float2* w, Body&& body)
{
    constexpr int NG = (N + G - 1) / G;                             // groups; the last one may be short
    const unsigned base = lds_addr(w);
    v2f ra[G], rb[G];
#pragma unroll
    for (int k = 0; k < G; ++k) ra[k] = rb[k] = v2f{0.f, 0.f};
    auto issue = [&](auto gc, v2f (&r)[G]) {
        constexpr int g = decltype(gc)::value;
        static_for<0, G>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = g * G + k;
            if constexpr (i < N) r[k] = lds_issue_b64<8 * (N - 1 - i)>(base);
        });
    };
    auto consume = [&](auto gc, v2f (&r)[G]) {
        constexpr int g = decltype(gc)::value;
        constexpr int next = (g + 1 < NG) ? ((g + 2) * G <= N ? G : N - (g + 1) * G) : 0;     // reads of group g + 1, issued after group g
        lds_landed_group<G, next>(r);
        static_for<0, G>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = g * G + k;
            if constexpr (i < N) body(icst<N - 1 - i>{}, r[k]);
        });
    };
    issue(icst<0>{}, ra);
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g % 2 == 0) {
            if constexpr (g + 1 < NG) issue(icst<g + 1>{}, rb);
            consume(gc, ra);
        } else {
            if constexpr (g + 1 < NG) issue(icst<g + 1>{}, ra);
            consume(gc, rb);
        }
    });
}

// window reads of the two FIRs: reads in flight ahead of their use (0 = leave the schedule to the compiler)
#ifndef P25FE_K1_LDS_DEPTH
#define P25FE_K1_LDS_DEPTH 8
#endif
// > 0: reads in groups of this many with one s_waitcnt per group (takes precedence over P25FE_K1_LDS_DEPTH)
#ifndef P25FE_K1_LDS_GROUP
#define P25FE_K1_LDS_GROUP 8
#endif
// body(j, w[j]) for j = N-1 .. 0 in the form the build selects
// (GROUPED = false: the kernels with caller-supplied taps, whose broadcast tap reads from LDS interleave with the
// window reads anyway and whose register budget -- 147 VGPRs -- has no room for the second group)
template <int N, bool GROUPED, class Body> __device__ __forceinline__ void lds_walk(const float2* w, Body&& body)
{
#if P25FE_K1_LDS_GROUP > 0
    if constexpr (GROUPED) {
        lds_walk_down_grouped<N, P25FE_K1_LDS_GROUP>(w, body);
        return;
    }
#endif
#if P25FE_K1_LDS_DEPTH > 0
    lds_walk_down<N, P25FE_K1_LDS_DEPTH>(w, body);
#else
    static_for<0, N>([&](auto ic) {
        constexpr int j = N - 1 - decltype(ic)::value;
        body(icst<j>{}, lds_read_v2(w + j));       // compiler-scheduled (volatile) reads
    });
#endif
}

// Value held by lane-1 of the wave; lane 0 receives `lane0` (DPP wave_shr:1 -- one VALU op, no LDS).
__device__ __forceinline__ float wave_shr1(float v, float lane0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// Wave-uniform copy of lane L's value (v_readlane_b32 -> SGPR).
template <int L> __device__ __forceinline__ float lane_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
}

// Lanes 0..9 of the result hold the ten wave-uniform words: lane q = low half of m[q], lane q + 5 = high half
// (v_writelane_b32; this hipcc has no builtin for it).  gfx950 needs 2 wait states between a VALU write of an SGPR
// (the v_cmp of a ballot) and a VALU read of it, and the compiler cannot see into the asm: all ten words are inputs of
// ONE block that starts with the padding.  (Without it the word of plane 0 -- v_cmp vcc immediately followed by the
// read of vcc_lo -- came out stale.)
__device__ __forceinline__ unsigned lanes_from_ballots(const unsigned long long (&m)[5])
{
    unsigned v = 0u;
    asm("s_nop 1\n\t"
        "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 5\n\t"
        "v_writelane_b32 %0, %3, 1\n\tv_writelane_b32 %0, %4, 6\n\t"
        "v_writelane_b32 %0, %5, 2\n\tv_writelane_b32 %0, %6, 7\n\t"
        "v_writelane_b32 %0, %7, 3\n\tv_writelane_b32 %0, %8, 8\n\t"
        "v_writelane_b32 %0, %9, 4\n\tv_writelane_b32 %0, %10, 9"
        : "+v"(v)
        : "s"((unsigned)m[0]), "s"((unsigned)(m[0] >> 32)), "s"((unsigned)m[1]), "s"((unsigned)(m[1] >> 32)),
          "s"((unsigned)m[2]), "s"((unsigned)(m[2] >> 32)), "s"((unsigned)m[3]), "s"((unsigned)(m[3] >> 32)),
          "s"((unsigned)m[4]), "s"((unsigned)(m[4] >> 32)));
    return v;
}
// value of v held by lane `src` (ds_bpermute_b32: the LDS crossbar, no memory)
__device__ __forceinline__ unsigned lane_gather(unsigned v, int src)
{
    return (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}

// Phase boundary inside the one-wave workgroup: LDS operations of one wave execute in order, so only
// the compiler has to be kept from moving accesses across the boundary.
__device__ __forceinline__ void phase_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SPEC 3.1: rtlsdr_iq LUT value as arithmetic (src/demod.rs:82-84) -- the immediate-coefficient kernels; the generic
// kernels (and a specialised build whose table is not affine) look the byte up in a 256-entry table in LDS
__device__ __forceinline__ float u8_to_f32(unsigned b) { return __builtin_fmaf((float)b, K1_CT_U8_SCALE, K1_CT_U8_OFFSET); }

// ------------------------------------------------------------------------------------------
// window loader: global -> registers, one 16-B vector per lane per load.
//
// Every load is unconditional and branch-free (bounds-checked buffer loads, see Loader::init); samples outside
// [-n_hist, n_new) that a straddling vector still brought in are zeroed in LDS by fixup(), which only boundary
// windows enter.  (An earlier version had a second, element-wise path for straddling vectors; two paths writing the
// same VGPRs made the compiler put s_waitcnt vmcnt(0) in front of every load, serialising 13 HBM round trips per
// sub-tile.)
// ------------------------------------------------------------------------------------------
// cache policy bits of the window loads (buffer-load aux: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef P25FE_K1_LD_AUX
#define P25FE_K1_LD_AUX 0
#endif
template <int FMT, int PK, int TX = 0, int NVX = 0> struct Loader {
    using G = Geo<PK, TX>;
    // A lane's vector always holds TWO samples -- 16 B of cf32 or 4 B of u8 pairs -- which become one 16-B LDS store,
    // lane-consecutive and therefore conflict-free.  (u8 used to load 16 B = 8 samples per lane: the 64-B lane stride of
    // the resulting ds_write_b128 is a 4-way bank conflict in every 8-lane group and cost 20 % of the kernel.)
    static constexpr int LOG_SPV = 1;
    static constexpr int SPV = 1 << LOG_SPV;
    static constexpr int NV = NVX ? NVX : (G::XWIN + SPV + SPV * WV - 1) / (SPV * WV);   // vectors per lane: 13 for PK = 5 (NVX: the segment prologue's short window)
    using V = typename cond<FMT == P25FE_FMT_CF32, uint4, unsigned>::type;
    V v[NV];

    // Window loads are BUFFER loads through one descriptor per segment: the hardware bounds check returns zeros for
    // every dword outside [descriptor base, base + num_records) and touches no memory for it, which replaces what
    // round 1 did per vector and per sub-tile in the vector ALU (clamp the vector index with v_max / v_min, sign-extend,
    // 64-bit add: 52 instructions for 13 loads) by ONE 32-bit add per load whose offset does not fit the 12-bit immediate.
    //   descriptor base  = the later of (first vector of the segment's first window, vector holding sample -n_hist)
    //   num_records      = bytes up to the last sample the segment needs (or the stream's end) -- so the prefetch past the
    //                      segment's end costs no traffic, as the clamp onto one cached vector did;
    //   a window that starts before the base has a negative byte offset = a huge unsigned one = out of range = zeros.
    // (An odd n_hist leaves one invalid sample inside the base vector: fixup() zeroes it, as it does every other sample
    // outside [-n_hist, n_new) that a straddling vector brought in.)
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    static constexpr int BPS = FMT == P25FE_FMT_CF32 ? 8 : 2;       // bytes per sample
    static constexpr int VB = BPS * SPV;                            // bytes per vector: 16 / 4
    __amdgpu_buffer_rsrc_t rs;
    long base_idx;                                                  // sample index of the descriptor base (uniform, vector aligned)

    // base: pointer to owned sample 0 of this channel; first0: first sample of the segment's first window;
    // i_last: last sample index this workgroup will ever need
    __device__ __forceinline__ void init(const void* base, long first0, long n_hist, long n_new, long i_last)
    {
        const long f_al = (first0 >> LOG_SPV) << LOG_SPV;
        const long lo_al = ((-n_hist) >> LOG_SPV) << LOG_SPV;
        base_idx = f_al > lo_al ? f_al : lo_al;
        const long last = i_last < n_new - 1 ? i_last : n_new - 1;
        // whole vectors: the range check works on the access (a u8 dword holding `last` and one sample more would be
        // rejected as a whole); the extra sample shares its vector, hence its page, with a valid one, and fixup() zeroes
        // it where it lies past the stream's end
        long bytes = ((last + 1 - base_idx + SPV - 1) >> LOG_SPV) * VB;
        bytes = bytes < 0 ? 0 : (bytes > (1L << 30) ? (1L << 30) : bytes);
        const uintptr_t p = reinterpret_cast<uintptr_t>(base) + (uintptr_t)(base_idx * BPS);
        // the descriptor must be provably wave-uniform (otherwise every load becomes a waterfall loop)
        const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)p), phi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)phi << 32) | plo), 0,
                                               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    }
    // The segment's FIRST window may start before the descriptor base (stream start): its lanes' byte offsets are
    // negative there.  The range check does not wrap a negative register offset plus a positive instruction offset
    // back into range (it sees a huge unsigned value even when the sum is a valid byte), so this one load computes every
    // offset in full and sends the negative ones far out of range.  Later windows of a segment start at or after the base.
    __device__ __forceinline__ void load_first(long first, int tid)
    {
        const long f_al = (first >> LOG_SPV) << LOG_SPV;
        long rel = (f_al - base_idx) * BPS;
        rel = rel < -(1L << 30) ? -(1L << 30) : (rel > (1L << 30) ? (1L << 30) : rel);
        const int voff = (int)rel + VB * tid;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int o = voff + j * VB * WV;
            o = o < 0 ? 0x7ffffff0 : o;
            if constexpr (FMT == P25FE_FMT_CF32) {
                const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, P25FE_K1_LD_AUX);
                v[j] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, P25FE_K1_LD_AUX);
            }
        }
    }
    __device__ __forceinline__ void load(long first, int tid)        // first >= base_idx (every window but a segment's first)
    {
        const long f_al = (first >> LOG_SPV) << LOG_SPV;
        long rel = (f_al - base_idx) * BPS;                         // uniform; saturate so that the 32-bit offset cannot wrap back in range
        rel = rel < -(1L << 30) ? -(1L << 30) : (rel > (1L << 30) ? (1L << 30) : rel);
        const int voff = (int)rel + VB * tid;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int o = P25FE_M_LOAD_OFFSET(voff + j * VB * WV, VB * tid);
            if constexpr (FMT == P25FE_FMT_CF32) {
                const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, P25FE_K1_LD_AUX);
                v[j] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, P25FE_K1_LD_AUX);
            }
        }
    }

    // XIN[k] receives sample first_al + k, first_al = first rounded down to a vector boundary: vector j of lane tid is
    // ONE aligned 16-B LDS store (ds_write_b128, lane-consecutive: conflict-free).  The consumers add the 0 / 1 sample
    // shift to their read base.  (Storing at XIN - shift made the alignment unknown at compile time: every staging
    // store became a ds_write2_b64 whose 16-B lane stride is a 2-way bank conflict -- 23 % of K1's LDS cycles.)
    // The stores are UNCONDITIONAL; samples outside [-n_hist, n_new) are zeroed afterwards by fixup(), which only
    // boundary windows enter.  (Round 1 masked the values before storing, under `if (!interior)`: the compiler
    // if-converted that into selects that every window executed -- 26 x (64-bit add, two 64-bit compares, two
    // v_cndmask) = ~150 of the 900 VALU instructions per sub-tile.  Conditional LDS stores cannot be speculated, so
    // the branch around fixup() survives.)
    template <bool LUTM = false>
    __device__ __forceinline__ void store(float2* XIN, long first, long n_hist, long n_new, int tid, const float* lut = nullptr) const
    {
        float4* XV = reinterpret_cast<float4*>(__builtin_assume_aligned(XIN, 16));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float2 s[SPV];
#pragma unroll
            for (int e = 0; e < SPV; ++e) {
                if constexpr (FMT == P25FE_FMT_CF32) {
                    const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                    s[e] = make_float2(__uint_as_float(w[2 * e]), __uint_as_float(w[2 * e + 1]));
                } else {
                    const unsigned pair = (v[j] >> (16 * e)) & 0xffffu;
                    if constexpr (LUTM) s[e] = make_float2(lut[pair & 0xffu], lut[pair >> 8]);
                    else s[e] = make_float2(u8_to_f32(pair & 0xffu), u8_to_f32(pair >> 8));   // low byte = I (SPEC 3.1)
                }
            }
            // only the last round of vectors can run past the window: everything else is stored unconditionally
            if (j < NV - 1 || SPV * (tid + j * WV) + SPV <= G::XIN_N) XV[tid + j * WV] = make_float4(s[0].x, s[0].y, s[1].x, s[1].y);
        }
    }
    __device__ __forceinline__ void fixup(float2* XIN, long first, long n_hist, long n_new, int tid) const
    {
        const long first_al = (first >> LOG_SPV) << LOG_SPV;
        if (first_al >= -n_hist && first_al + (long)NV * SPV * WV <= n_new) return;      // uniform: interior window
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < SPV; ++e) {
                const long i = first_al + SPV * (long)(tid + j * WV) + e;
                const int k = SPV * (tid + j * WV) + e;
                if ((i < -n_hist || i >= n_new) && k < G::XIN_N) XIN[k] = make_float2(0.f, 0.f);
            }
    }
};

// ------------------------------------------------------------------------------------------
// K1: fused front end.  One wave walks `subs_per_seg` consecutive sub-tiles of one channel.
//   * the next sub-tile's input window is prefetched into registers while the current one is
//     processed; it reaches LDS (time-linear cf32) at the top of the next iteration;
//   * both FIRs are register-tiled: a lane computes PK consecutive outputs from one sliding
//     window of ds_read_b64 (tap order and single accumulator as SPEC 3.2 prescribes);
//   * FM discriminator and boxcar never touch LDS: a lane has its own PK channel outputs in
//     registers and gets its left neighbours' values with DPP wave_shr:1; what crosses a sub-tile
//     boundary is carried in SGPRs (v_readlane of lanes 63, 62, ...);
//   * the d history ((T2-1) samples) is carried in LDS, so the filter halo is recomputed once
//     per segment only.
//
// Local indexing: owned input sample i in [0, n_new); output m' in [0, n_out) is produced by
// input i = o0 + 5 m' (o0 = first decimation instant inside the owned range, SPEC 3.2).
// ------------------------------------------------------------------------------------------
struct K1Args {
    const void* x;          // owned sample 0 of channel 0
    long ch_stride;         // in samples
    long n_hist;            // valid samples before x (per channel)
    long n_new;             // owned samples (per channel)
    int o0;                 // 0..4
    float* bb;              // baseband out, channel 0
    long bb_stride;
    long n_out;             // outputs per channel
    int subs_per_seg;
    int seg_count;          // segments per channel in this launch; work items = seg_count x n_ch, channel-major
    int n_ch;               // channels
    int seg_first;          // first segment of this launch (a shard's head segment is launched after its halo has arrived)
    long m_begin;           // first output to produce (<= 0: also outputs that lie in the history)
    float* power_partial;   // nullable: [n_channels][seg_count] partial sums of |y|^2
    // OUT_PLANAR only (bb unused): polyphase baseband + sign planes, see PLPAD.  Prologue form: m_begin + pl_shift >= 0 and a
    // multiple of 320 (every sub-tile is one block of the layout); halo form: m_begin + pl_shift >= 0, a multiple
    // of 80, and a segment length that is a multiple of 80.
    float* bbp;             // channel 0
    long bbp_ch_stride;     // floats per channel (a whole number of blocks)
    uint8_t* bits;          // channel 0, addressed by byte
    long bits_ch_stride;    // bytes per channel
    int pl_shift;           // output m sits at planar position m + pl_shift: PLPAD, or PLPAD + 2 when the receiver runs with
                            // the tracking clock's lookahead (p25fe_recv.hip)
    // OUT_PLANAR, nullable: the launch's LAST workgroup to finish publishes done_flag[0] = done_seq (agent-scope release;
    // done_flag[1] is the ticket counter, zero between launches).  A time shard's head segment runs on another stream than
    // the sync detection that reads its planes: the detection's first tile polls this word (p25fe_recv.hip, k_detect).
    unsigned* done_flag;
    unsigned done_seq;
    // The first lead_segs segments of the range are ONE sub-tile long (the rest subs_per_seg): a time shard's head -- the
    // outputs that depend on the halo, launched on their own after it has arrived -- is then two or three one-sub-tile
    // workgroups side by side (~6 us) instead of one workgroup walking a whole segment alone (15 - 45 us).
    int lead_segs;
    int seg_halo;           // generic kernels, halo form: outputs a segment recomputes in front of its first one (seg_halo_for; the
                            // immediate-coefficient kernels know theirs at compile time)
};

// CT = true: the handle's taps are the build's default tables (p25fe_spec.h) -> immediates, no
// registers or LDS reads spent on coefficients.  CT = false: caller-supplied taps, broadcast-read from LDS.
// Register budget of the planar variants: minimum waves per SIMD the kernel is compiled for.  cf32 (halo form): bound 3
// and bound 2 are both allocated ~150 VGPRs, but the bound-3 schedule ran 300 us against 272 (round 2, same box): 2.  u8
// (prologue form): 148 VGPRs either way; 3.  (The prologue form of the cf32 kernel takes 174 VGPRs unbounded = 2 waves
// per SIMD, 168 with two 8-byte spills outside the sub-tile loop when bound to 3: 0.264 / 0.259 ms against 0.253.)
#ifndef P25FE_K1_PLANAR_WPS
#define P25FE_K1_PLANAR_WPS 2
#endif
#ifndef P25FE_K1_PLANAR_WPS_U8
#define P25FE_K1_PLANAR_WPS_U8 3
#endif
// Windows in flight per wave (register-staged loader).  Measured (tools/ab.sh, same box): a second register set costs
// the cf32 kernel a wave per SIMD (181 VGPRs -> 8 waves per CU instead of 11) and makes it 25 % SLOWER (318 vs 254 us)
// although 45 % more bytes are in flight -- the waves' arithmetic phases, not the bytes in flight, are what covers the
// latency there; the u8 kernel keeps 4 waves per SIMD with two sets (13 more VGPRs) and gains 10 % (200 vs 221 us).
#ifndef P25FE_K1_TURNAROUND_PRIO
#define P25FE_K1_TURNAROUND_PRIO 3
#endif
#ifndef P25FE_K1_PF_CF32
#define P25FE_K1_PF_CF32 1
#endif
// planar baseband rows as non-temporal stores: the 115 MB of a pass are read back sparsely (one plane in ten by K4, a
// few windows by K2) and only evict the window stream from L2 / Infinity Cache on their way out.  Round 2, first half:
// "-1 % on K1, +3 % on K4: not adopted"; with the receive kernels overlapped by the next K1 the trade is the other way
// round: pipelined step 0.270 / 0.270 / 0.275 against 0.278 / 0.283 / 0.282 ms (three interleaved bench runs, one box),
// K1 0.259 against 0.259 - 0.270, K4 0.0155 against 0.0150.
#ifndef P25FE_K1_NT_STORES
#define P25FE_K1_NT_STORES 1
#endif
#ifndef P25FE_K1_PF_U8
#define P25FE_K1_PF_U8 2
#endif
template <int FMT, bool CT, int PK, int OM = OUT_LINEAR, int TX = 0>
__device__ __forceinline__ void frontend_body(const K1Args& a, const Taps* __restrict__ gtaps)
{
    constexpr int PF = (PK != 5 || TX != 0) ? 1 : (FMT == P25FE_FMT_U8 ? P25FE_K1_PF_U8 : P25FE_K1_PF_CF32);
#ifndef P25FE_JIT
    static_assert(TX == 0 || !CT, "the 64-tap geometry is for caller-supplied taps");
#endif
    static_assert(OM == OUT_LINEAR || PK == 5, "the planar epilogue maps a 320-sample sub-tile onto 10 planes x 32 symbols");
    P25FE_M_ENTRY;
    using G = Geo<PK, TX>;
    constexpr int SUB = G::SUB;
    constexpr int P = PK;
    constexpr int T1 = G::T1, T2 = G::T2, D_CARRY = G::D_CARRY;      // this instantiation's tap counts (shadow the build's)
    // Post-discriminator filter (src/demod.rs:31, 52, 114; docs/SPEC.md 3.5).  T3 = its length as far as the GEOMETRY goes: the
    // handle's own for the immediate-coefficient kernels, the ABI's ceiling for the generic ones (n_avg at run time).  Up to
    // AVG_DPP_MAX taps it runs in registers (DPP, SGPR carries); longer filters through an LDS window of fm values.
    constexpr int T3 = CT ? K1_CT_AVG_N : TMAX;
    constexpr bool AVG_DPP = CT && T3 <= AVG_DPP_MAX && !(P25FE_K1_AVG_LDS_CF32 != 0 && FMT == P25FE_FMT_CF32);
    constexpr int HY = T3;                                          // channel outputs needed in front of a segment's first output
    constexpr int NBACK = AVG_DPP ? ((T3 - 1 + PK - 1) / PK > 0 ? (T3 - 1 + PK - 1) / PK : 1) : 1;   // lanes to the left whose fm values the filter needs
    static_assert(T3 >= 1 && T3 <= TMAX && NBACK <= 3, "post-discriminator filter length");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* D = reinterpret_cast<float2*>(smem);                    // [D_CARRY | SUB]: the SUB part aliases the window region (register loader)
    float2* XIN = D + D_CARRY;                                      // 16-B aligned: staged with ds_write_b128
    float* OUT = reinterpret_cast<float*>(D + G::D_N);              // [SUB] output transpose: inside the window region
    float* TAPS = reinterpret_cast<float*>(D + D_CARRY + G::XIN_N); // generic kernels only: [T1 | T2 | 3 | post-discriminator TMAX + 1]
    float* const LUT = TAPS + (CT ? 0 : G::TAPS_N);                 // [256] u8 table, when LUTM
    float* const AVT = TAPS + (T1 + T2 + 3);                        // generic kernels: the post-discriminator taps
    // u8 -> float: arithmetic with immediate constants (CT, affine table) or a table in LDS (the generic kernels always:
    // one code path for every table; a specialised build only when its table is not affine)
    constexpr bool LUTM = FMT == P25FE_FMT_U8 && (!CT || K1_CT_U8_LUT != 0);
    const int tid = threadIdx.x;
    if (!CT) {
        for (int k = tid; k < T1; k += WV) TAPS[k] = gtaps->dec[k];
        for (int k = tid; k < T2; k += WV) TAPS[T1 + k] = gtaps->ch[k];
        for (int k = tid; k < TMAX; k += WV) AVT[k] = gtaps->avg[k];
    }
    if constexpr (LUTM)
        for (int k = tid; k < 256; k += WV) LUT[k] = gtaps->lut[k];
    float* const FM = LUT + (FMT == P25FE_FMT_U8 && (!CT || K1_CT_U8_LUT != 0) ? 256 : 0);   // [T3 - 1 | SUB] fm window, when !AVG_DPP
    float fm_gain = K1_CT_FM_GAIN;
    int n_avg = T3, avg_uniform = K1_CT_AVG_UNIFORM;               // run-time values of the generic kernels
    if constexpr (!CT) {
        fm_gain = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gtaps->fm_gain)));
        n_avg = __builtin_amdgcn_readfirstlane(gtaps->n_avg);
        avg_uniform = __builtin_amdgcn_readfirstlane(gtaps->avg_uniform);
    }
    // segment halo of the halo form (cf32): 80 outputs, 160 when the two filters behind the decimator remember more than that
    const int SEGH = CT ? seg_halo_for(T3, T2) : a.seg_halo;
    auto tap_dec = [&](int k) -> float { return CT ? K1_CT_DECIM_TAPS[k] : TAPS[k]; };
    auto tap_ch = [&](int k) -> float { return CT ? K1_CT_CHAN_TAPS[k] : TAPS[T1 + k]; };

    constexpr bool PRO = seg_prologue(FMT);                         // segment prologue (u8) or recomputed halo (cf32), see SEG_HALO
    const long seg_len = PRO ? (long)a.subs_per_seg * SUB : (long)(SUB - SEGH) + (long)(a.subs_per_seg - 1) * SUB;
    // segment prologue: the ND decimator outputs in front of the segment, from a window of PWIN input samples
    constexpr int ND = HY + (T2 - 1);                               // 50 with the build's numbers
    constexpr int PWIN = DEC * (ND - 1) + T1;                       // 276 (424)
    constexpr int NVP = (PWIN + 2 + 2 * WV - 1) / (2 * WV);         // 16-B vectors per lane: 3 (4)
    constexpr int PBASE = 2 * WV * NVP;                             // the d's are parked behind the staged prologue window
    constexpr int NDL = (ND + WV - 1) / WV;                         // d's per lane: 1 (2)
    static_assert(PBASE + ND <= G::XIN_N && D_CARRY <= ND, "prologue scratch fits the window region");
    P25FE_M_LOCALS;
    // Work items (segment, channel) are taken in a grid-stride loop: the launch holds one workgroup per resident wave
    // slot, wave b works on items b, b + G, b + 2 G, ... (consecutive items = consecutive segments of one channel, so at
    // any moment the resident waves stream one compact, advancing region of the capture -- what the dispatcher's in-order
    // hand-out of 3-sub-tile workgroups gave, without the ~4 us a wave slot stood empty between two workgroups).
    // (The default launch is one workgroup per item on a (segment, channel) grid: no division for the channel index.)
    const long n_items = (long)a.seg_count * a.n_ch;
    const bool grid2d = gridDim.x == (unsigned)a.seg_count;         // uniform: one workgroup per item, blockIdx.y = channel
    for (long item = grid2d ? (long)blockIdx.y * a.seg_count + blockIdx.x : (long)blockIdx.x; item < n_items;
         item += grid2d ? n_items : (long)gridDim.x) {
    P25FE_M_ITEM;
    const int ch = grid2d ? (int)blockIdx.y : (int)((unsigned)item / (unsigned)a.seg_count);
    const long seg_rel = grid2d ? (long)blockIdx.x : item - (long)ch * a.seg_count;
    const long seg = seg_rel + a.seg_first;
    const bool lead = seg < (long)a.lead_segs;                      // uniform
    const long LEAD_LEN = PRO ? (long)SUB : (long)(SUB - SEGH);
    const long this_len = lead ? LEAD_LEN : seg_len;
    const int subs_this = lead ? 1 : a.subs_per_seg;
    const long m_seg0 = a.m_begin + (lead ? seg * LEAD_LEN : (long)a.lead_segs * LEAD_LEN + (seg - a.lead_segs) * seg_len);
    if (m_seg0 >= a.n_out) continue;
    const long m_seg1 = (m_seg0 + this_len < a.n_out) ? m_seg0 + this_len : a.n_out;
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)ch * a.ch_stride * (FMT == P25FE_FMT_CF32 ? 8 : 2);
    float* bb = a.bb + (size_t)ch * a.bb_stride;
    phase_sync();                                                   // the previous item's LDS reads precede this item's writes

    using LoaderT = Loader<FMT, PK, TX>;
    using LoaderP = Loader<FMT, PK, TX, NVP>;
    LoaderT ld0, ld1;
    long dlo = PRO ? m_seg0 : m_seg0 - SEGH;                       // first d index of this sub-tile
    const long i_last = (long)a.o0 + DEC * (m_seg1 - 1);          // newest input sample this segment needs
    float pw = 0.f;
    // context carried across sub-tiles in wave-uniform registers: the last channel output and the
    // fm values of the last NBACK lanes of the previous sub-tile
    float2 y_carry = make_float2(0.f, 0.f);
    float f_carry[NBACK][P];
    if constexpr (!PRO) {
        // halo form: zero carries; the first sub-tile's first SEGH outputs absorb the filters' start-up and are dropped
        for (int k = tid; k < D_CARRY; k += WV) D[k] = make_float2(0.f, 0.f);
        if constexpr (!AVG_DPP)
            for (int k = tid; k < T3 - 1; k += WV) FM[k] = 0.f;
        ld0.init(xb, (long)a.o0 + DEC * dlo - (T1 - 1), a.n_hist, a.n_new, i_last);
        ld0.load_first((long)a.o0 + DEC * dlo - (T1 - 1), tid);
        if constexpr (PF == 2) {
            ld1.rs = ld0.rs; ld1.base_idx = ld0.base_idx;
            ld1.load((long)a.o0 + DEC * (dlo + SUB) - (T1 - 1), tid);
        }
#pragma unroll
        for (int b = 0; b < NBACK; ++b)
#pragma unroll
            for (int p = 0; p < P; ++p) f_carry[b][p] = 0.f;
    } else {
    LoaderP lp;
    const long pfirst = (long)a.o0 + DEC * (m_seg0 - ND) - (T1 - 1);     // first input sample of the prologue window
    ld0.init(xb, pfirst, a.n_hist, a.n_new, i_last);
    lp.rs = ld0.rs; lp.base_idx = ld0.base_idx;
    // one HBM round trip for the prologue window and the first sub-tile's window (both may start before the descriptor's base
    // at the start of a stream: load_first sends those lanes out of range = zeros)
    lp.load_first(pfirst, tid);
    // cf32 (16-byte vectors: 52 registers per window): the first sub-tile's window is requested only once the prologue window
    // has been staged and its registers are free again -- otherwise the two in flight together set the kernel's register count
    // (174: two waves per SIMD); the prologue's arithmetic runs under the second request instead of under both.
    constexpr bool LATE_LD0 = FMT == P25FE_FMT_CF32;
    if constexpr (!LATE_LD0) {
        ld0.load_first((long)a.o0 + DEC * dlo - (T1 - 1), tid);
        if constexpr (PF == 2) {
            ld1.rs = ld0.rs; ld1.base_idx = ld0.base_idx;
            ld1.load_first((long)a.o0 + DEC * (dlo + SUB) - (T1 - 1), tid);
        }
    }
    {
        // ---- prologue: d[m_seg0 - ND .. m_seg0) on lanes 0 .. ND-1, y[m_seg0 - 10 .. m_seg0) and the discriminator values
        // behind them on lanes 0..9 -- the state a receiver that had run through the previous segment would hand over.
        // Same tap order and single accumulators as the sub-tiles (SPEC 3.2 - 3.4): the same bits.
        float2* const PD_ = XIN + PBASE;
        lp.template store<LUTM>(XIN, pfirst, a.n_hist, a.n_new, tid, LUT);
        lp.fixup(XIN, pfirst, a.n_hist, a.n_new, tid);
        if constexpr (LATE_LD0) {
            ld0.load_first((long)a.o0 + DEC * dlo - (T1 - 1), tid);
            if constexpr (PF == 2) {
                ld1.rs = ld0.rs; ld1.base_idx = ld0.base_idx;
                ld1.load_first((long)a.o0 + DEC * (dlo + SUB) - (T1 - 1), tid);
            }
        }
        phase_sync();
        const int pxsh = (int)(pfirst & 1);
        v2f dacc[NDL];
#pragma unroll
        for (int r = 0; r < NDL; ++r) {
            dacc[r] = v2f{0.f, 0.f};
            const int l = tid + WV * r;
            if (NDL == 1 || l < ND) {                               // (one d per lane: the 14 surplus lanes compute on staged zeros / garbage, never stored)
                const float2* w = XIN + pxsh + DEC * l;             // lane stride 5 complex = 10 dwords: conflict-free ds_read_b64
                lds_walk<T1, CT>(w, [&](auto jc, v2f sv) {           // newest to oldest => tap order 0..T1-1
                    constexpr int j = decltype(jc)::value;
                    dacc[r] = cfma(tap_dec(T1 - 1 - j), sv, dacc[r]);
                });
            }
        }
#pragma unroll
        for (int r = 0; r < NDL; ++r) {
            const int l = tid + WV * r;
            if (l < ND) PD_[l] = make_float2(dacc[r].x, dacc[r].y);
        }
        phase_sync();
        v2f yv = v2f{0.f, 0.f};
        if (tid < HY) {
            const float2* w = PD_ + (ND - HY) - (T2 - 1) + tid;     // d[m - k] = w[T2 - 1 - k] for m = m_seg0 - HY + tid
            lds_walk<T2, CT>(w, [&](auto jc, v2f sv) {
                constexpr int j = decltype(jc)::value;
                yv = cfma(tap_ch(T2 - 1 - j), sv, yv);
            });
        }
        float2 yp;
        yp.x = wave_shr1(yv.x, 0.f);
        yp.y = wave_shr1(yv.y, 0.f);
        const float fmv = fm_discriminate(make_float2(yv.x, yv.y), yp, fm_gain);     // lane l: fm[m_seg0 - HY + l], l = 1..HY-1 (lane 0: unused)
        y_carry.x = lane_bcast<HY - 1>(yv.x);
        y_carry.y = lane_bcast<HY - 1>(yv.y);
        static_for<0, NBACK>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            static_for<0, P>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                constexpr int l = HY - (b + 1) * P + p;              // f_carry[b][p] = fm of lane 63 - b, output p, of the sub-tile before
                if constexpr (AVG_DPP && l >= 1) f_carry[b][p] = lane_bcast<l>(fmv);
                else f_carry[b][p] = 0.f;                            // further back than the filter reaches
            });
        });
        if constexpr (!AVG_DPP)
            if (tid >= 1 && tid < HY) FM[tid - 1] = fmv;             // the fm window's history: fm[m_seg0 - (T3 - 1) .. m_seg0)
        for (int k = tid; k < D_CARRY; k += WV) D[k] = PD_[ND - D_CARRY + k];
        phase_sync();
    }
    }   // PRO

    // Outputs are kept in registers for one sub-tile and stored at the top of the next one, BEFORE the
    // prefetch loads are issued: vmcnt counts loads and stores in one in-order queue, so stores issued
    // after the prefetch would force the wait at the top of the loop to drain them too (measured: the
    // loop then ran at the HBM write round-trip per sub-tile).
    float outv[P];
#pragma unroll
    for (int q = 0; q < P; ++q) outv[q] = 0.f;
    int out_rel = -2 * SUB;                                         // out_lo - m_seg0; nothing in range yet
    const int seg_n = (int)(m_seg1 - m_seg0);
    float* const bb_seg = OM == OUT_LINEAR ? bb + m_seg0 : nullptr;
    // planar: lane = (half h = tid >> 5, symbol tid & 31) holds, in outv[q], plane 5 h + q of the sub-tile's 32 symbols
    const int pl_sym = tid & 31, pl_h5 = (tid >> 5) * 5;
    const int i_seg = OM == OUT_PLANAR ? (int)((m_seg0 + a.pl_shift) / SPS_) : 0;     // symbol index of the segment's first output (multiple of 32)
    float* const bbp_ch = OM == OUT_PLANAR ? a.bbp + (size_t)ch * a.bbp_ch_stride : nullptr;
    uint8_t* const bits_ch = OM == OUT_PLANAR ? a.bits + (size_t)ch * a.bits_ch_stride : nullptr;
    unsigned bitsv = 0u;                                            // prologue form: lanes 0..9 hold the sign word of plane tid; halo form: lanes 0..39 byte (tid & 3) of plane (tid >> 2)
    auto flush_outputs = [&]() {
        if constexpr (OM == OUT_LINEAR) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int r = out_rel + tid + q * WV;               // output index relative to the segment start
                if (r >= 0 && r < seg_n) bb_seg[r] = outv[q];
            }
        } else {
            if constexpr (PRO) {
            // a sub-tile is one block of the layout: ten 128-byte rows and ten sign words
            const int blk = (i_seg >> 5) + out_rel / SUB;
            float* const row = bbp_ch + (size_t)blk * PL_BLK + pl_sym + 32 * pl_h5;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int r = out_rel + SPS_ * pl_sym + pl_h5 + q;
#if P25FE_K1_NT_STORES
                if (r >= 0 && r < seg_n) __builtin_nontemporal_store(outv[q], row + 32 * q);
#else
                if (r >= 0 && r < seg_n) row[32 * q] = outv[q];
#endif
            }
            // lane j < 10 holds plane j's 32 sign bits of the sub-tile (the range's last word may carry bits past n_out: unread)
            if (tid < SPS_ && out_rel >= 0 && out_rel < seg_n)
                reinterpret_cast<unsigned*>(bits_ch)[(size_t)blk * SPS_ + tid] = bitsv;
            } else {
            const int i_sub = i_seg + out_rel / SPS_;               // symbol index of the sub-tile (multiple of 8; negative only for
                                                                    // recomputed-halo outputs in front of the range, which r >= 0 below keeps from being stored)
            const int i = i_sub + pl_sym;
            float* const row = bbp_ch + (size_t)(i >> 5) * PL_BLK + (i & 31) + 32 * pl_h5;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int r = out_rel + SPS_ * pl_sym + pl_h5 + q;
                if constexpr (P25FE_M_NO_BB_STORES) { P25FE_M_KEEP2(outv[q], r); }
                else {
#if P25FE_K1_NT_STORES
                if (r >= 0 && r < seg_n) __builtin_nontemporal_store(outv[q], row + 32 * q);
#else
                if (r >= 0 && r < seg_n) row[32 * q] = outv[q];
#endif
                }
            }
            // a byte = 8 symbols = 80 consecutive outputs of one plane; out_rel is a multiple of 80, so a byte is
            // either wholly inside the segment or wholly halo (the range's last byte may carry bits past n_out: unread)
            const int r0 = out_rel + 8 * SPS_ * (tid & 3) + (tid >> 2);
            const int ib = i_sub + 8 * (tid & 3);
            if constexpr (!P25FE_M_NO_SIGN_PLANES)
            if (tid < 4 * SPS_ && r0 >= 0 && r0 < seg_n)
                bits_ch[((size_t)(ib >> 5) * SPS_ + (tid >> 2)) * 4 + ((ib >> 3) & 3)] = (uint8_t)bitsv;
            }
        }
    };

    // One sub-tile.  `ld` holds its window (requested PF sub-tiles ago) and is refilled with the window PF sub-tiles ahead.
    auto sub_tile = [&](auto& ld) -> bool {
        if (dlo >= m_seg1) return false;                           // uniform
        P25FE_M_SUBTILE;
        const long first = (long)a.o0 + DEC * dlo - (T1 - 1);      // XIN[xsh + k] = x[first + k]
        const int xsh = (int)(first & 1);                           // window start relative to the aligned staging origin
#if P25FE_K1_TURNAROUND_PRIO
        __builtin_amdgcn_s_setprio(P25FE_K1_TURNAROUND_PRIO);       // see below
#endif
        ld.template store<LUTM>(XIN, first, a.n_hist, a.n_new, tid, LUT);
        ld.fixup(XIN, first, a.n_hist, a.n_new, tid);
        phase_sync();
        K1_STAMP(0);                                                // window landed + staged
        flush_outputs();                                            // previous sub-tile's outputs (lane-predicated)
        // unconditional prefetch: past the segment's end the clamp makes every lane read one cached vector
        ld.load(first + (long)(PF * DEC) * SUB, tid);
#if P25FE_K1_TURNAROUND_PRIO
        // The turnaround -- window landed -> staged to LDS -> next window requested -- is the only part of the iteration during
        // which this wave has nothing in flight; it runs at raised priority so that the other waves' FMA streams do not
        // stretch it.  (Idle phases of the length of the arithmetic do not slow the load stream at all: 189 vs 192 us in a
        // loads-only build with s_sleep in place of the arithmetic.)
        __builtin_amdgcn_s_setprio(0);
#endif
        K1_STAMP(1);                                                // previous outputs stored, next window requested

        P25FE_M_CUT_LOADS;                                          // (measurement builds: stop after the load pipeline)
        // ---- stage 2: 5:1 decimating FIR (src/demod.rs:87). Lane: d[dlo + P tid + p], p = 0..P-1.
        // Output p needs x[first + 5(P tid + p) + (T1-1) - k], k = 0..T1-1  -> XIN[5 P tid + 5p + 30 - k].
        {
            const float2* w = XIN + xsh + (DEC * P) * tid;
            v2f acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = v2f{0.f, 0.f};
            lds_walk<DEC * (P - 1) + T1, CT>(w, [&](auto jc, v2f s) {   // newest to oldest => tap order 0..T1-1
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = DEC * p + (T1 - 1) - j;
                    if (k >= 0 && k < T1) acc[p] = cfma(tap_dec(k), s, acc[p]);
                }
            });
            // d overwrites the front of the window: every lane's window reads must be complete first.  (The compiler
            // reasons per thread and could prove a lane's own store and loads disjoint -- the fence orders the wave.)
            phase_sync();
            P25FE_M_PIN_V2(acc, P);
            K1_STAMP(2);                                            // decimator
#pragma unroll
            for (int p = 0; p < P; ++p) D[D_CARRY + P * tid + p] = make_float2(acc[p].x, acc[p].y);
        }
        phase_sync();

        P25FE_M_CUT(2);
        // ---- stage 3: channel FIR (src/demod.rs:93). Lane: y[dlo + P tid + p] from D[P tid + p + 40 - k].
        float2 y[P];
        {
            const float2* w = D + (D_CARRY - (T2 - 1)) + P * tid;    // d[m - k] sits at D[D_CARRY + (m - dlo) - k]
            v2f yv[P];
#pragma unroll
            for (int p = 0; p < P; ++p) yv[p] = v2f{0.f, 0.f};
            lds_walk<(P - 1) + T2, CT>(w, [&](auto jc, v2f s) {
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = p + (T2 - 1) - j;
                    if constexpr (P25FE_M_NO_CH_FMA) { if (k == 0) yv[p] = s; }
                    else if (k >= 0 && k < T2) yv[p] = cfma(tap_ch(k), s, yv[p]);
                }
            });
#pragma unroll
            for (int p = 0; p < P; ++p) y[p] = make_float2(yv[p].x, yv[p].y);
            if (a.power_partial) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const long m = dlo + P * tid + p;
                    if (m >= m_seg0 && m < m_seg1 && m >= 0) {
                        const float q0 = y[p].x * y[p].x, q1 = y[p].y * y[p].y;
                        pw = pw + (q0 + q1);
                    }
                }
            }
        }
        P25FE_M_PIN_F2(y, P);
        K1_STAMP(3);                                                // d stores + channel filter
        phase_sync();                                               // all reads of D done before its carry is rewritten
        // carry the d history to the next sub-tile (ordered before its next use by the next phase boundaries)
        for (int k = tid; k < D_CARRY; k += WV) D[k] = D[SUB + k];

        P25FE_M_CUT_KEEP_F2(3, y, P);
        // ---- stage 4: FM discriminator (src/demod.rs:109-111) on the lane's own P outputs; the sample before
        // the first one comes from lane-1 (DPP), for lane 0 from the previous sub-tile (SGPR carry)
        float f[P];
        {
            float2 yprev;
            yprev.x = wave_shr1(y[P - 1].x, y_carry.x);
            yprev.y = wave_shr1(y[P - 1].y, y_carry.y);
            f[0] = fm_discriminate(y[0], yprev, fm_gain);
#pragma unroll
            for (int p = 1; p < P; ++p) f[p] = fm_discriminate(y[p], y[p - 1], fm_gain);
            y_carry.x = lane_bcast<WV - 1>(y[P - 1].x);
            y_carry.y = lane_bcast<WV - 1>(y[P - 1].y);
        }
        K1_PIN2(f[0], f[1]); K1_PIN2(f[2], f[P - 2]); K1_PIN2(f[P - 1], f[0]);
        K1_STAMP(4);                                                // d carry copy + discriminator
        P25FE_M_CUT_KEEP_F(4, f, P);
        // ---- stage 5: post-discriminator filter (src/demod.rs:114; docs/SPEC.md 3.5).  The reference's is MovingAverage::new(10):
        // b[m] = (fm[m] + fm[m-1] + ... + fm[m-9]) * 0.1, summed newest first -- the rule for every table whose taps are all
        // equal; any other table is a FIR like the two before it: acc = +0, acc = fma(h[k], fm[m - k], acc) in tap order.
        if constexpr (AVG_DPP) {
        // fm[P tid + p - j] lives in this lane (q = p - j >= 0) or in lane - b, b = ceil(-q / P): prevf[b-1][q + b P].
        {
            float prevf[NBACK][P];
#pragma unroll
            for (int p = 0; p < P; ++p) prevf[0][p] = wave_shr1(f[p], f_carry[0][p]);
#pragma unroll
            for (int b = 1; b < NBACK; ++b)
#pragma unroll
                for (int p = 0; p < P; ++p) prevf[b][p] = wave_shr1(prevf[b - 1][p], f_carry[b][p]);
            auto fm_back = [&](int p, int j) -> float {              // fm[P tid + p - j]
                const int q = p - j;
                if (q >= 0) return f[q];
                const int b = (-q + P - 1) / P;                     // 1 .. NBACK
                return prevf[b - 1][q + b * P];
            };
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float acc;
                if constexpr (K1_CT_AVG_UNIFORM) {
                    acc = f[p];
#pragma unroll
                    for (int j = 1; j < T3; ++j) acc = acc + fm_back(p, j);
                    acc = acc * K1_CT_AVG_TAPS[0];
                } else {
                    acc = 0.f;
#pragma unroll
                    for (int j = 0; j < T3; ++j) acc = __builtin_fmaf(K1_CT_AVG_TAPS[j], fm_back(p, j), acc);
                }
                if constexpr (P25FE_M_NO_OUT_STAGE) { P25FE_M_KEEP1(acc); }
                else OUT[P * tid + p] = acc;                        // lane stride P dwords (odd): conflict-free
            }
            // next sub-tile's lane 0 / 1 / ... read these: lane 63 is one lane back, lane 62 two, ...
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (NBACK >= 3) f_carry[2][p] = lane_bcast<WV - 3>(f[p]);
                if constexpr (NBACK >= 2) f_carry[1][p] = lane_bcast<WV - 2>(f[p]);
                f_carry[0][p] = lane_bcast<WV - 1>(f[p]);
            }
        }
        } else {
            // longer filters (and the generic kernels, whose length is a run-time number): the sub-tile's fm values go behind
            // the T3 - 1 carried ones in LDS, every lane walks the window of its P outputs (lane stride P dwords: conflict-free)
#pragma unroll
            for (int p = 0; p < P; ++p) FM[(T3 - 1) + P * tid + p] = f[p];
            phase_sync();
            const float* w = FM + (T3 - 1) + P * tid;                // w[p - k] = fm[m - k] for the lane's output p
            float acc[P];
            if constexpr (CT) {
                // one window value per step: v = fm[m0 + P - 1 - i] feeds output p at tap j = p + i - (P - 1); a scheduling
                // barrier every 8 steps keeps the compiler from hoisting all T3 + P - 1 reads (and their registers) to the top
                if constexpr (K1_CT_AVG_UNIFORM) {
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = 0.f;
                    static_for<0, T3 + P - 1>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const float v = w[(P - 1) - i];
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const int j = p + i - (P - 1);
                            if (j == 0) acc[p] = v;                 // the newest sample starts the sum (SPEC 3.5: no +0 in front)
                            else if (j > 0 && j < T3) acc[p] = acc[p] + v;
                        }
                        if constexpr (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                    });
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = acc[p] * K1_CT_AVG_TAPS[0];
                } else {
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = 0.f;
                    static_for<0, T3 + P - 1>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const float v = w[(P - 1) - i];
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const int j = p + i - (P - 1);
                            if (j >= 0 && j < T3) acc[p] = __builtin_fmaf(K1_CT_AVG_TAPS[j], v, acc[p]);
                        }
                        if constexpr (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                    });
                }
            } else if (avg_uniform) {                               // uniform: the handle's numbers
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = w[p];
                for (int j = 1; j < n_avg; ++j)
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = acc[p] + w[p - j];
                const float sc = AVT[0];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = acc[p] * sc;
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = 0.f;
                for (int j = 0; j < n_avg; ++j) {
                    const float hj = AVT[j];
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = __builtin_fmaf(hj, w[p - j], acc[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) OUT[P * tid + p] = acc[p];
            phase_sync();                                           // every lane's window reads precede the carry's rewrite
            for (int k = tid; k < T3 - 1; k += WV) FM[k] = FM[SUB + k];
        }
        if constexpr (P25FE_M_NO_OUT_STAGE) return true;
        phase_sync();
        K1_STAMP(5);                                                // boxcar + transpose stores
        // transpose through LDS: lane-consecutive outputs -> coalesced (deferred) global stores
        if constexpr (OM == OUT_LINEAR) {
#pragma unroll
            for (int q = 0; q < P; ++q) outv[q] = OUT[tid + q * WV];
        } else {
            // polyphase: lanes 0..31 take planes 0..4, lanes 32..63 planes 5..9 of the sub-tile's 32 symbols (lane stride 10
            // dwords: a 2-way conflict on 5 reads); a ballot of the sign bits is one 32-symbol word of two planes at once.
            unsigned w = 0u;
            unsigned long long sg[5];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                outv[q] = OUT[SPS_ * pl_sym + pl_h5 + q];
                sg[q] = __builtin_amdgcn_ballot_w64(__float_as_int(outv[q]) < 0);
            }
            if constexpr (!P25FE_M_NO_SIGN_PLANES) w = lanes_from_ballots(sg);        // lane q: word of plane q, lane q + 5: plane q + 5
            if constexpr (PRO) {
                bitsv = w;
            } else {
                const unsigned pw_ = lane_gather(w, tid >> 2);      // lane j <- word of plane j >> 2
                bitsv = (pw_ >> (8 * (tid & 3))) & 0xffu;
            }
        }
        out_rel = (int)(dlo - m_seg0);
        phase_sync();
        K1_PIN2(outv[0], outv[P - 1]);
        K1_STAMP(6);                                                // transpose reads (+ sign words)
        return true;
    };
    if constexpr (PF == 1) {
        for (int it = 0; it < subs_this; ++it, dlo += SUB)
            if (!sub_tile(ld0)) break;
    } else {
        // two register sets, two windows in flight per wave: the loop is unrolled by two so that each set keeps its registers
        for (int it = 0; it < subs_this; it += 2) {
            if (!sub_tile(ld0)) break;
            dlo += SUB;
            if (it + 1 >= subs_this || !sub_tile(ld1)) break;
            dlo += SUB;
        }
    }
    flush_outputs();

    if (a.power_partial) {
        // wave reduction of the |y|^2 partials (tree; tolerance vs the sequential fold is in the tests)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) pw = pw + __shfl_down(pw, d, 64);
        if (tid == 0) a.power_partial[(size_t)ch * a.seg_count + seg_rel] = pw;     // any order: summed by k_power_finish
    }
    }   // work items
    if constexpr (OM == OUT_PLANAR) {
        if (a.done_flag) {                                          // uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the compiler may drop the wait behind the write-back
            unsigned ticket = 0u;
            if (tid == 0) ticket = __hip_atomic_fetch_add(a.done_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
            if (ticket == gridDim.x * gridDim.y - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (tid == 0) {
                    __hip_atomic_store(a.done_flag + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    P25FE_M_EXIT;
}

// (register budget of the generic u8 kernels: one wave per SIMD less than the immediate-coefficient ones -- the taps' LDS
// addresses and the table look-ups cost registers, and a spilled FIR loop costs far more than the occupancy)
constexpr int k1_wps(int fmt, bool ct, int pk, int om)
{
    if (om == OUT_PLANAR) return fmt == P25FE_FMT_U8 ? (ct ? P25FE_K1_PLANAR_WPS_U8 : 2) : P25FE_K1_PLANAR_WPS;
    return (fmt == P25FE_FMT_U8 && !ct && pk <= 3) ? 3 : (pk <= 3 ? 4 : 2);
}
template <int FMT, bool CT, int PK, int OM = OUT_LINEAR, int TX = 0>
__global__ __launch_bounds__(WV, k1_wps(FMT, CT, PK, OM)) void k_frontend(K1Args a, const Taps* __restrict__ gtaps)
{
    frontend_body<FMT, CT, PK, OM, TX>(a, gtaps);
}

// ------------------------------------------------------------------------------------------
// K0: stage 0 of BASELINE.json config 3 -- 10:1 decimating FIR, 2.4 Msps cf32 -> 240 ksps cf32, T0 = 80 taps
// (SPEC 3.0; no reference counterpart: src/consts.rs:11 fixes 240 ksps).  Same construction as K1: one wave per
// workgroup, next window prefetched into registers, outputs stored one iteration late.  The 1990-sample window of
// a 192-output sub-tile is staged in LDS in POLYPHASE layout X[r][j] = x[base + 10 j + r]; a lane computes 3
// consecutive outputs i = 3 lane + p: tap k = 10 q + r' needs X[9 - r'][3 lane + u] with u = p + 7 - q, so walking
// u = 9..0 and loading the ten phases of column 3 lane + u feeds up to three accumulators, each in tap order
// 0..79 (lane stride 3 complex = 6 dwords: conflict-free ds_read_b64; 100 reads for 240 packed FMAs).
// A pure FIR carries no state between sub-tiles: the 70-sample left context is simply part of every window.
// Algorithmic bytes: 8 B read + 0.8 B written per input sample.
// ------------------------------------------------------------------------------------------
constexpr int PD = P25FE_PRE_DECIM;          // 10
constexpr int T0 = P25FE_T0;                 // 80
#ifndef P25FE_K0_P
#define P25FE_K0_P 3
#endif
constexpr int K0_P = P25FE_K0_P;             // outputs per lane (odd)
constexpr int K0_SUB = WV * K0_P;            // 192 outputs per sub-tile
#ifndef P25FE_K0_SUBS
#define P25FE_K0_SUBS 2
#endif
#ifndef P25FE_K0_WPS
#define P25FE_K0_WPS 2
#endif
constexpr int K0_SUBS = P25FE_K0_SUBS;       // sub-tiles per workgroup (prefetch pipeline depth)
constexpr int K0_HALO = T0 - PD;             // 70 input samples of left context
constexpr int K0_NIN = PD * K0_SUB + K0_HALO;       // 1990 window positions
constexpr int k0_row_pitch(int need) { return need + ((13 - need % 16) + 16) % 16; }
constexpr int K0_JP = k0_row_pitch(K0_NIN / PD + 2);   // entries per phase row (205 for 192 outputs); = 13 mod 16: staging position t lands on
                                             // bank pair 13 t + c (mod 16) whether or not the phase wraps -> conflict-free ds_write_b64
static_assert(K0_JP >= K0_NIN / PD + 2 && K0_JP % 16 == 13, "polyphase row pitch");
constexpr int K0_NV = (K0_NIN + 2 + 2 * WV - 1) / (2 * WV);   // 16-B vectors per lane: 16
constexpr int HIST_PRE = K0_HALO + PD - 1;   // 79: history needed for exact results

struct K0Args {
    const float* x;         // owned sample 0 of channel 0 (cf32 @ 2.4 Msps), 16-B aligned
    long ch_stride;         // samples
    long n_hist, n_new;
    int o0;                 // first decimation instant inside the owned range, 0..9
    float* y;               // cf32 out @ 240 ksps, channel 0
    long y_stride;          // samples
    long n_out;
};

#ifndef P25FE_JIT
__global__ __launch_bounds__(WV, P25FE_K0_WPS) void k_predecim(K0Args a)
{
    __shared__ float2 X[PD * K0_JP];
    __shared__ float2 OUT[K0_SUB];
    const int tid = threadIdx.x, ch = blockIdx.y;
    const long m_wg0 = (long)blockIdx.x * (K0_SUB * K0_SUBS);
    const uint4* xb = reinterpret_cast<const uint4*>(a.x + 2 * (size_t)ch * a.ch_stride);
    float2* yb = reinterpret_cast<float2*>(a.y) + (size_t)ch * a.y_stride;

    const long wg_last = (long)a.o0 + PD * (m_wg0 + (long)(K0_SUBS - 1) * K0_SUB) - (T0 - 1);   // base of the last window
    uint4 v[K0_NV];
    // window of the sub-tile whose first output is m0: positions k = 0 .. K0_NIN-1 are inputs base + k
    auto load = [&](long m0) {
        const long base = (long)a.o0 + PD * m0 - (T0 - 1);
        const long v0 = base >> 1;
        long lo = ((-a.n_hist) >> 1) - v0, hi = ((a.n_new - 1) >> 1) - v0;
        const long seg_hi = ((wg_last + K0_NIN) >> 1) - v0;         // prefetch past the segment re-reads its last vector
        hi = hi < seg_hi ? hi : seg_hi;
        lo = lo < -(1L << 30) ? -(1L << 30) : (lo > (1L << 30) ? (1L << 30) : lo);
        hi = hi < -(1L << 30) ? -(1L << 30) : (hi > (1L << 30) ? (1L << 30) : hi);
        const int lo32 = (int)lo, hi32 = (int)hi;
        const uint4* q = xb + v0;
#pragma unroll
        for (int j = 0; j < K0_NV; ++j) {
            int r = tid + j * WV;
            r = r < lo32 ? lo32 : r;
            r = r > hi32 ? hi32 : r;
            v[j] = q[r];
        }
    };
    auto stage = [&](long m0) {
        const long base = (long)a.o0 + PD * m0 - (T0 - 1);
        const long v0 = base >> 1;
        const int sh = (int)(base - (v0 << 1));                     // 0 or 1
        const bool interior = (v0 << 1) >= -a.n_hist && (v0 << 1) + 2L * K0_NV * WV <= a.n_new;   // uniform
#pragma unroll
        for (int j = 0; j < K0_NV; ++j) {
            const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * (tid + j * WV) + e - sh;          // window position
                float2 s2 = make_float2(__uint_as_float(w[2 * e]), __uint_as_float(w[2 * e + 1]));
                if (!interior) {
                    const long i = ((v0 + tid + (long)j * WV) << 1) + e;
                    if (i < -a.n_hist || i >= a.n_new) s2 = make_float2(0.f, 0.f);
                }
                if (k >= 0 && k < K0_NIN) X[((unsigned)k % PD) * K0_JP + (unsigned)k / PD] = s2;
            }
        }
    };

    float2 outv[K0_P];
#pragma unroll
    for (int q = 0; q < K0_P; ++q) outv[q] = make_float2(0.f, 0.f);
    long out_m0 = -1;                                               // sub-tile whose outputs sit in outv (-1: none)
    auto flush = [&]() {
        if (out_m0 >= 0) {
#pragma unroll
            for (int q = 0; q < K0_P; ++q) {
                const long m = out_m0 + tid + q * WV;
                if (m < a.n_out) yb[m] = outv[q];
            }
        }
    };

    load(m_wg0);
#pragma unroll 1
    for (int it = 0; it < K0_SUBS; ++it) {
        const long m0 = m_wg0 + (long)it * K0_SUB;
        if (m0 >= a.n_out) break;                                   // uniform
        stage(m0);
        phase_sync();
        flush();                                                    // stores before the prefetch (single in-order vmcnt queue)
        load(m0 + K0_SUB);                                          // unconditional; clamped past the stream
        float2 acc[K0_P];
#pragma unroll
        for (int p = 0; p < K0_P; ++p) acc[p] = make_float2(0.f, 0.f);
        const float2* col = X + K0_P * tid;
#pragma unroll
        for (int u = K0_P - 1 + T0 / PD - 1; u >= 0; --u) {        // u = 9 .. 0
            float2 ph[PD];
#pragma unroll
            for (int r = 0; r < PD; ++r) ph[r] = lds_read_c(col + r * K0_JP + u);
#pragma unroll
            for (int p = 0; p < K0_P; ++p) {
                const int q = p + (T0 / PD - 1) - u;                // tap block 10 q .. 10 q + 9
                if (q >= 0 && q < T0 / PD) {
#pragma unroll
                    for (int rr = 0; rr < PD; ++rr) {
                        const float h = P25FE_DEFAULT_PRE_TAPS[PD * q + rr];
                        acc[p].x = __builtin_fmaf(h, ph[PD - 1 - rr].x, acc[p].x);
                        acc[p].y = __builtin_fmaf(h, ph[PD - 1 - rr].y, acc[p].y);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < K0_P; ++p) OUT[K0_P * tid + p] = acc[p];
        phase_sync();
#pragma unroll
        for (int q = 0; q < K0_P; ++q) outv[q] = OUT[tid + q * WV];
        out_m0 = m0;
        phase_sync();
    }
    flush();
}
#endif

// ------------------------------------------------------------------------------------------
// K6: polyphase channeliser (SPEC 3.11; SURVEY.md section 8f rank 4, no reference counterpart -- the reference tunes
// one channel, src/sdr.rs:64-65).  One 2.4 Msps capture -> 192 channels on the 12.5 kHz raster, each at 240 ksps, i.e.
// the stream stage 1 would deliver for a tuner on that channel:
//     y_c[m] = sum_{p<80} h[p] x[n-p] e^{-j 2 pi c (n-p)/192}
//            = e^{-j 2 pi c n/192} * sum_p (h[p] x[n-p]) e^{+j 2 pi c p/192},        n = absolute instant, h = SPEC 3.0's taps.
// The inner sum is a 192-point DFT of 80 non-zero inputs.  With p = 16 p1 + p2 and c = c1 + 12 c2 the kernel
// e^{+j 2 pi c p/192} splits into V12^{c1 p1} * V192^{c1 p2} * V16^{c2 p2} (the cross term is a whole turn):
//   for each c1 (12, uniform loop):  B[p2] = (sum_{p1<5} a[16 p1 + p2] V12^{c1 p1}) * V192^{c1 p2}      (pruned radix-12)
//                                    Y[c2] = sum_{p2<16} B[p2] V16^{c2 p2}                               (radix-4 x 4 FFT)
//                                    y[c1 + 12 c2] = Y[c2] * conj(V192^{(c n) mod 192})                  (rotation, LDS table)
// ~6.5 kFLOP per output instant instead of 61 k for the direct contraction: this is why the stage is NOT cast as a
// GEMM for the matrix cores -- in fp32 they run at the vector rate and the dense form costs 9x the arithmetic, while
// the kernel is bound by writing 192 x 8 B per 10 input samples (161.6 B per input sample, 154 of them output).
// A lane owns one output instant (all twiddles are compile-time immediates, every store is 512 contiguous bytes of
// one channel row); the 80 products a[p] stay in registers, the window is staged polyphase as in K0.
// ------------------------------------------------------------------------------------------
constexpr int CZ_M = P25FE_CHZ_CHANNELS;                        // 192
constexpr int CZ_C1 = 12, CZ_C2 = 16, CZ_P1 = T0 / CZ_C2;       // c = c1 + 12 c2; p = 16 p1 + p2, p1 < 5
static_assert(T0 == CZ_C2 * CZ_P1 && CZ_M == CZ_C1 * CZ_C2 && CZ_P1 == 5, "factorisation of the 192-point DFT");
// c1 values whose partial sums are held at once.  Round 4, same box, three interleaved rounds (tools/k6_ab.sh, profiles/
// r04_k6_k0_ab.txt): 4 -> 4.91 - 4.93 ms with 168 VGPRs + 96 B / lane of scratch in the twiddle loop; 3 -> 4.43 ms, 157 VGPRs,
// no scratch (one more pass over the 80 window products per instant, which the stores hide); 2 -> 4.48 ms.
#ifndef P25FE_CZ_G
#define P25FE_CZ_G 3
#endif
#ifndef P25FE_CZ_WPS
#define P25FE_CZ_WPS 3
#endif
constexpr int CZ_G = P25FE_CZ_G;                                         // c1 values whose partial sums are held at once
constexpr int CZ_NIN = PD * WV + (T0 - PD);                     // 710 window positions for 64 instants
constexpr int CZ_JP = 77;                                       // entries per phase row (>= 73), = 13 mod 16 as in K0
static_assert(CZ_JP >= CZ_NIN / PD + 2 && CZ_JP % 16 == 13, "polyphase row pitch");
constexpr int CZ_NV = (CZ_NIN + 2 + 2 * WV - 1) / (2 * WV);     // 16-B vectors per lane: 6

struct ChzArgs {
    const float* x;         // owned sample 0 (cf32 @ 2.4 Msps), 16-B aligned
    long n_hist, n_new;
    long abs0;              // absolute index of owned sample 0
    int o0;                 // first decimation instant inside the owned range, 0..9
    float* y;               // [192][y_stride] cf32 @ 240 ksps
    long y_stride;          // samples
    long n_out;
};

__device__ __forceinline__ float2 cz_mul_const(float2 v, float wr, float wi)      // v * (wr + j wi), constants folded
{
    if (wi == 0.0f) return wr == 1.0f ? v : make_float2(v.x * wr, v.y * wr);
    if (wr == 0.0f) return make_float2(-(v.y * wi), v.x * wi);
    return make_float2(__builtin_fmaf(-v.y, wi, v.x * wr), __builtin_fmaf(v.y, wr, v.x * wi));
}
// c1-dependent twiddles, read with scalar loads inside the (rolled) c1 loop: [c1][p1 - 1] = V12^{c1 p1} for p1 = 1..4,
// [c1][4 + p2] = V192^{c1 p2}
struct CzC { float x, y; };
struct CzTw { CzC v[CZ_C1][4 + CZ_C2]; };
constexpr CzTw cz_make_tw()
{
    CzTw t{};
    for (int c1 = 0; c1 < CZ_C1; ++c1) {
        for (int p1 = 1; p1 < CZ_P1; ++p1) {
            const int k = 16 * ((c1 * p1) % CZ_C1);
            t.v[c1][p1 - 1].x = P25FE_CHZ_W[2 * k]; t.v[c1][p1 - 1].y = P25FE_CHZ_W[2 * k + 1];
        }
        for (int p2 = 0; p2 < CZ_C2; ++p2) {
            const int k = c1 * p2;
            t.v[c1][4 + p2].x = P25FE_CHZ_W[2 * k]; t.v[c1][4 + p2].y = P25FE_CHZ_W[2 * k + 1];
        }
    }
    return t;
}
#ifndef P25FE_JIT
__constant__ CzTw CZ_TW = cz_make_tw();
#endif
__device__ __forceinline__ float2 cz_mac(float2 acc, float2 v, CzC w)              // acc + v * w
{
    acc.x = __builtin_fmaf(v.x, w.x, acc.x); acc.y = __builtin_fmaf(v.y, w.x, acc.y);
    acc.x = __builtin_fmaf(-v.y, w.y, acc.x); acc.y = __builtin_fmaf(v.x, w.y, acc.y);
    return acc;
}
__device__ __forceinline__ float2 cz_mul(float2 v, CzC w)
{
    return make_float2(__builtin_fmaf(-v.y, w.y, v.x * w.x), __builtin_fmaf(v.y, w.x, v.x * w.y));
}
// radix-4 butterfly for the +j kernel: y_f = sum_a x_a (+j)^{f a}
__device__ __forceinline__ void cz_bf4(float2& x0, float2& x1, float2& x2, float2& x3)
{
    const float2 t0 = make_float2(x0.x + x2.x, x0.y + x2.y), t1 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 t2 = make_float2(x1.x + x3.x, x1.y + x3.y), t3 = make_float2(x1.x - x3.x, x1.y - x3.y);
    x0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    x2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    x1 = make_float2(t1.x - t3.y, t1.y + t3.x);                 // t1 + j t3
    x3 = make_float2(t1.x + t3.y, t1.y - t3.x);                 // t1 - j t3
}

#ifndef P25FE_JIT
__global__ __launch_bounds__(WV, P25FE_CZ_WPS) void k_channelise(ChzArgs a)
{
    __shared__ float2 X[PD * CZ_JP];
    __shared__ float2 ROT[CZ_M];
    const int lane = threadIdx.x;
    for (int k = lane; k < CZ_M; k += WV) ROT[k] = make_float2(P25FE_CHZ_W[2 * k], -P25FE_CHZ_W[2 * k + 1]);
    const long m0 = (long)blockIdx.x * WV;
    {   // stage the window: positions k = 0 .. CZ_NIN-1 are inputs base + k
        const uint4* xb = reinterpret_cast<const uint4*>(a.x);
        const long base = (long)a.o0 + PD * m0 - (T0 - 1);
        const long v0 = base >> 1;
        const int sh = (int)(base - (v0 << 1));
        long lo = ((-a.n_hist) >> 1) - v0, hi = ((a.n_new - 1) >> 1) - v0;
        lo = lo < -(1L << 30) ? -(1L << 30) : (lo > (1L << 30) ? (1L << 30) : lo);
        hi = hi < -(1L << 30) ? -(1L << 30) : (hi > (1L << 30) ? (1L << 30) : hi);
        const int lo32 = (int)lo, hi32 = (int)hi;
        const uint4* q = xb + v0;
        uint4 v[CZ_NV];
#pragma unroll
        for (int j = 0; j < CZ_NV; ++j) {
            int r = lane + j * WV;
            r = r < lo32 ? lo32 : r;
            r = r > hi32 ? hi32 : r;
            v[j] = q[r];
        }
        const bool interior = (v0 << 1) >= -a.n_hist && (v0 << 1) + 2L * CZ_NV * WV <= a.n_new;   // uniform
#pragma unroll
        for (int j = 0; j < CZ_NV; ++j) {
            const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * (lane + j * WV) + e - sh;
                float2 s2 = make_float2(__uint_as_float(w[2 * e]), __uint_as_float(w[2 * e + 1]));
                if (!interior) {
                    const long i = ((v0 + lane + (long)j * WV) << 1) + e;
                    if (i < -a.n_hist || i >= a.n_new) s2 = make_float2(0.f, 0.f);
                }
                if (k >= 0 && k < CZ_NIN) X[((unsigned)k % PD) * CZ_JP + (unsigned)k / PD] = s2;
            }
        }
    }
    phase_sync();

    const long m = m0 + lane;
    const unsigned nm = (unsigned)((unsigned long long)(a.abs0 + a.o0 + PD * m) % (unsigned long long)CZ_M);
    const unsigned step2 = (CZ_C1 * nm) % (unsigned)CZ_M;       // index step of the rotation from c to c + 12
    unsigned idx1 = 0;                                          // (c1 * nm) mod 192
    float2* yb = reinterpret_cast<float2*>(a.y) + m;
    const bool odd = (lane & 1) != 0;

    // c1 in groups of CZ_G (a rolled loop: the c1-dependent twiddles come from CZ_TW by scalar loads): the 80 products
    // a[p] = h[p] x[n - p] are re-formed from LDS once per group (window position 79 + 10 lane - p -> phase
    // 9 - p % 10, column lane + 7 - p / 10) instead of living in 160 registers
    P25FE_M_K6_STORES_ONLY;                                         // (measurement builds: the store stream alone)
#pragma unroll 1
    for (int c1g = 0; c1g < CZ_C1; c1g += CZ_G) {
        float2 BB[CZ_G][CZ_C2];
#pragma unroll
        for (int p2 = 0; p2 < CZ_C2; ++p2) {
            float2 a5[CZ_P1];
#pragma unroll
            for (int p1 = 0; p1 < CZ_P1; ++p1) {
                const int p = CZ_C2 * p1 + p2;
                const float2 xv = lds_read_c(X + (PD - 1 - p % PD) * CZ_JP + lane + (T0 / PD - 1) - p / PD);
                a5[p1] = make_float2(P25FE_DEFAULT_PRE_TAPS[p] * xv.x, P25FE_DEFAULT_PRE_TAPS[p] * xv.y);
            }
#pragma unroll
            for (int q = 0; q < CZ_G; ++q) {
                const CzC* tw = CZ_TW.v[c1g + q];
                float2 acc = a5[0];                             // p1 = 0: V12^0 = 1
#pragma unroll
                for (int p1 = 1; p1 < CZ_P1; ++p1) acc = cz_mac(acc, a5[p1], tw[p1 - 1]);
                float2 bv = p2 ? cz_mul(acc, tw[4 + p2]) : acc;
                // pin the value here: otherwise the arithmetic is sunk below ALL the (ordered, volatile) window reads of the
                // group and their 160 result registers stay live -- spills at any useful occupancy
                asm volatile("" : "+v"(bv.x), "+v"(bv.y));
                BB[q][p2] = bv;
            }
        }
#pragma unroll
        for (int q = 0; q < CZ_G; ++q) {
        const int c1 = c1g + q;
        // 16-point DFT, kernel V16^{c2 p2} = e^{+j 2 pi c2 p2 / 16}: p2 = 4 a + b, c2 = e + 4 f
#pragma unroll
        for (int b = 0; b < 4; ++b) cz_bf4(BB[q][b], BB[q][4 + b], BB[q][8 + b], BB[q][12 + b]);        // slot 4 e + b <- Z[b][e]
#pragma unroll
        for (int e = 1; e < 4; ++e)
#pragma unroll
            for (int b = 1; b < 4; ++b) {
                const int k = 12 * (e * b);                     // V16^{e b} = W[12 e b]
                BB[q][4 * e + b] = cz_mul_const(BB[q][4 * e + b], P25FE_CHZ_W[2 * k], P25FE_CHZ_W[2 * k + 1]);
            }
        unsigned idx[4];                                        // rotation index of c = c1 + 12 (e + 4 f), by e
        idx[0] = idx1;
#pragma unroll
        for (int e = 1; e < 4; ++e) { idx[e] = idx[e - 1] + step2; idx[e] = idx[e] >= (unsigned)CZ_M ? idx[e] - CZ_M : idx[e]; }
        const unsigned step8 = (4u * step2) % (unsigned)CZ_M;   // from f to f + 1: c grows by 48
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cz_bf4(BB[q][4 * e], BB[q][4 * e + 1], BB[q][4 * e + 2], BB[q][4 * e + 3]);                 // slot 4 e + f <- Y[e + 4 f]
            unsigned ix = idx[e];
            float2 o4[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float2 rot = ROT[ix];
                const float2 yv = BB[q][4 * e + f];
                o4[f] = make_float2(__builtin_fmaf(-yv.y, rot.y, yv.x * rot.x), __builtin_fmaf(yv.y, rot.x, yv.x * rot.y));
                ix += step8; ix = ix >= (unsigned)CZ_M ? ix - CZ_M : ix;
            }
            // Streaming stores (nt: 22 GB of output per minute of capture never fit a cache; measured +4.5 %), 16 bytes per
            // lane: the lanes of a pair swap one value each (DPP quad_perm [1,0,3,2]) so that the even lane writes instants
            // (m, m + 1) of channel row f and the odd lane instants (m - 1, m) of row f + 1 -- half as many store
            // instructions for the same 512 contiguous bytes per row (the store-only build: 0.416 against 0.458 ms per
            // 1.44e7 input samples).  Rows are padded to whole tiles: no predicate.
#pragma unroll
            for (int f = 0; f < 4; f += 2) {
                const float2 mine = odd ? o4[f + 1] : o4[f];        // what this lane keeps (its own instant)
                const float2 give = odd ? o4[f] : o4[f + 1];        // what its partner needs
                float2 got;
                got.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give.x), 0xB1, 0xf, 0xf, true));
                got.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give.y), 0xB1, 0xf, 0xf, true));
                typedef float __attribute__((ext_vector_type(4))) f32x4;
                f32x4 ov;
                ov.x = odd ? got.x : mine.x; ov.y = odd ? got.y : mine.y;      // instant m - 1 (odd lane) / m (even lane)
                ov.z = odd ? mine.x : got.x; ov.w = odd ? mine.y : got.y;      // instant m (odd lane) / m + 1 (even lane)
                const int c = c1 + CZ_C1 * (e + 4 * (f + (odd ? 1 : 0)));
                __builtin_nontemporal_store(ov, reinterpret_cast<f32x4*>(yb - (odd ? 1 : 0) + (size_t)c * a.y_stride));
            }
        }
        idx1 += nm; idx1 = idx1 >= (unsigned)CZ_M ? idx1 - CZ_M : idx1;
        }
    }
}
#endif

// finish power_dbm (src/demod.rs:123-134): 30 + 10 log10( (sum / N) / R ), R = 1
#ifndef P25FE_JIT
__global__ void k_power_finish(const float* partial, int n_partial, long n, float* out_dbm)
{
    const int ch = blockIdx.x;
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s = s + partial[(size_t)ch * n_partial + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float avg = red[0] / (float)n;
        out_dbm[ch] = 30.0f + 10.0f * log10f(avg / 1.0f);       // n = 0: 0 / 0 -> NaN, as the reference's fold over an empty chunk
    }
}
#endif

}  // namespace p25k

#include "p25fe_recv.hip"   // K2..K4: the symbol receiver (sync detect, scan, slicer) on the polyphase layout

namespace p25k {

// ------------------------------------------------------------------------------------------
// K5: network identifier after each frame sync (MessageEvent::PacketNID, src/recv.rs:216-222).
// One workgroup per sync event gathers the 32 NID dibits (status symbol at +11 skipped) and finds the nearest of the
// 65536 BCH(63,16,23) code words by exhaustive Hamming search (256 per thread, two 256-entry XOR tables in LDS) --
// equivalent to a bounded-distance decoder when at most t = 11 bits are wrong, and embarrassingly parallel.
// ------------------------------------------------------------------------------------------
// Batch form: blockIdx.y = channel, per-channel strides, event / dibit counts read from the channel's p25fe_result_t.
#ifndef P25FE_JIT
__global__ __launch_bounds__(256) void k_nid(const uint8_t* dibits, unsigned long long n_dibits, const unsigned long long* sync_dibit,
                                             const long* sync_pos, p25fe_nid_t* out, const p25fe_result_t* results,
                                             unsigned long long dibit_stride, unsigned long long sync_stride)
{
    __shared__ unsigned long long LO[256], HI[256];
    __shared__ unsigned long long raw_sh;
    __shared__ unsigned best_sh[4];
    const int tid = threadIdx.x, k = blockIdx.x;
    if (results) {                                              // uniform
        const p25fe_result_t r = results[blockIdx.y];
        if ((unsigned long long)k >= r.n_sync) return;
        n_dibits = r.n_dibits;
        dibits += blockIdx.y * dibit_stride;
        sync_dibit += blockIdx.y * sync_stride;
        if (sync_pos) sync_pos += blockIdx.y * sync_stride;
        out += blockIdx.y * sync_stride;
    }
    {
        unsigned long long a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (tid & (1 << i)) { a ^= P25FE_NID_ROWS[i]; b ^= P25FE_NID_ROWS[8 + i]; }
        LO[tid] = a; HI[tid] = b;
    }
    const unsigned long long D = sync_dibit[k];
    const bool have = D + 33 <= n_dibits;
    if (tid == 0) {
        unsigned long long raw = 0;
        if (have)
            for (int j = 0; j < 33; ++j)
                if (j != 11) raw = (raw << 2) | (unsigned long long)(dibits[D + j] & 3u);
        raw_sh = raw;
    }
    __syncthreads();
    const unsigned long long cw = raw_sh >> 1;
    unsigned best = 0xffffffffu;                                // (distance << 16) | data : min = nearest, smallest data on ties
    for (unsigned d = (unsigned)tid; d < 65536u; d += 256u) {
        const unsigned dist = (unsigned)__popcll((HI[d >> 8] ^ LO[d & 255u]) ^ cw);
        const unsigned key = (dist << 16) | d;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = __shfl_down(best, o, 64);
        best = v < best ? v : best;
    }
    if ((tid & 63) == 0) best_sh[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned b = best_sh[0];
        for (int w = 1; w < 4; ++w) b = best_sh[w] < b ? best_sh[w] : b;
        p25fe_nid_t r;
        r.raw = have ? raw_sh : 0ull;
        r.sync_pos = sync_pos ? sync_pos[k] : 0;
        const unsigned data = b & 0xffffu, dist = b >> 16;
        r.nac = have ? (uint16_t)(data >> 4) : 0;
        r.duid = have ? (uint8_t)(data & 15u) : 0;
        r.n_errors = have ? (uint8_t)dist : 0;
        r.valid = have ? (dist <= P25FE_NID_T ? 1 : 0) : -1;
        out[k] = r;
    }
}
#endif

// Per-channel observability record (src/hub.rs:344, 401-402, 557-581): one workgroup per
// channel folds the channel's NID records into the "bch" CodeStats row and copies the lock state.
#ifndef P25FE_JIT
__global__ __launch_bounds__(256) void k_chan_stats(const p25fe_result_t* results, const p25fe_nid_t* nid,
                                                    unsigned long long sync_stride, const float* power_dbm,
                                                    p25fe_chan_stats_t* stats)
{
    __shared__ unsigned long long sh[3][4];
    const int tid = threadIdx.x, ch = blockIdx.x;
    const p25fe_result_t r = results[ch];
    unsigned long long words = 0, errs = 0, fixed = 0;
    if (nid) {
        const unsigned long long n = r.n_sync < sync_stride ? r.n_sync : sync_stride;
        for (unsigned long long k = tid; k < n; k += 256) {
            const p25fe_nid_t v = nid[ch * sync_stride + k];
            if (v.valid >= 0) { ++words; if (v.valid == 0) ++errs; else fixed += v.n_errors; }
        }
    }
    unsigned long long acc[3] = {words, errs, fixed};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
        if ((tid & 63) == 0) sh[q][tid >> 6] = acc[q];
    }
    __syncthreads();
    if (tid == 0) {
        p25fe_chan_stats_t o;
        o.sig_power_dbm = power_dbm ? power_dbm[ch] : __int_as_float(0x7fc00000);
        o.locked = r.anchor_out.valid != 0 ? 1 : 0;                 // (the tracking clock keeps a fraction in valid's upper bits)
        o.n_dibits = r.n_dibits;
        o.n_sync = r.n_sync;
        o.last_sync_pos = (r.n_sync > 0 && r.anchor_out.valid) ? r.anchor_out.s : -1;
        o.bch.words = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        o.bch.errs = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        o.bch.fixed = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
        o.bch.size = 63; o.bch.reserved = 0;
        stats[ch] = o;
    }
}
#endif

// ------------------------------------------------------------------------------------------
// One launch per streaming chunk (the body of DemodTask::run + the sample loop of RecvTask::run for one buffer of the
// reader, src/demod.rs:70-117, src/recv.rs:145-150): K1's workgroups write the planes of a range of at most one tile; the
// LAST of a channel's workgroups to finish (agent-scope release, ticket, agent-scope acquire: cdna_hip_programming.md
// guideline 16) runs the receiver for that channel and writes the results where the host reads them.
// ------------------------------------------------------------------------------------------
struct ChunkTail {
    ChunkRecvArgs r;
    unsigned* counter;          // [ch], zero between launches
    int wg_per_ch;
};

template <int FMT, bool CT, int TX>
__device__ __forceinline__ void chunk_body(const K1Args& a, const Taps* __restrict__ gtaps, const ChunkTail& t)
{
    frontend_body<FMT, CT, 5, OUT_PLANAR, TX>(a, gtaps);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the compiler may drop the wait behind the write-back
    unsigned ticket = 0u;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&t.counter[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket != (unsigned)t.wg_per_ch - 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) t.counter[blockIdx.y] = 0u;               // the next launch (a kernel boundary later) starts from zero
    recv_one_tile(t.r, (int)blockIdx.y);
}

template <int FMT, bool CT, int TX>
__global__ __launch_bounds__(WV, 2) void k_chunk(K1Args a, const Taps* __restrict__ gtaps, ChunkTail t)
{
    chunk_body<FMT, CT, TX>(a, gtaps, t);
}

// (shard_resolve_impl, the carry resolution across time shards, lives in p25fe_recv.hip: the slicer runs it too)

// Dibit gather, second half: the all-gathered, padded per-shard streams -> one contiguous stream.  Shard r's dibits
// are gathered[r * cap .. + offset[r + 1] - offset[r]) and belong at out[offset[r] ..).
#ifndef P25FE_JIT
__global__ __launch_bounds__(256) void k_shard_compact(const uint8_t* gathered, unsigned long long cap, const uint64_t* offset,
                                                        int first_shard, int n_shards, uint8_t* out, unsigned long long out_cap)
{
    const int r = first_shard + (int)blockIdx.y;                   // (rank 0 slices its own shard straight into the stream: first_shard = 1)
    if (r >= n_shards) return;
    const unsigned long long o0 = offset[r], n = offset[r + 1] - o0;
    const uint8_t* src = gathered + (unsigned long long)r * cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n && i < cap; i += (unsigned long long)gridDim.x * 256)
        if (o0 + i < out_cap) out[o0 + i] = src[i];
}
#endif

#ifndef P25FE_JIT
// Do two streams share a hardware queue?  HIP multiplexes the streams of one priority level onto four queues; streams on one queue
// run their kernels strictly one after the other.  k_queue_probe_wait (stream A) waits for a word k_queue_probe_set (stream B, launched
// after it) writes -- for at most `ticks` of the 100 MHz wall clock: on one queue the setter cannot start before the waiter has given up.
__global__ void k_queue_probe_wait(unsigned* word, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned seen = 0u;
    while (!(seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    word[1] = seen ? 1u : 2u;                                        // 1: the other stream ran beside this one; 2: it did not
}
__global__ void k_queue_probe_set(unsigned* word) { __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

#ifndef P25FE_JIT
// anchor_out of C result records -> a contiguous anchor array (the carry-in of the next window of p25fe_run_host_windows)
__global__ void k_anchors_from_results(const p25fe_result_t* results, p25fe_anchor_t* anchors, int n_ch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_ch) anchors[c] = results[c].anchor_out;
}
#endif

#ifndef P25FE_JIT
__global__ void k_shard_resolve(const p25fe_result_t* summaries, const uint64_t* shard_bb0, const uint64_t* shard_bb_n,
                                int n_shards, int symbol_clock, p25fe_anchor_t* anchor_in, uint64_t* dibit_offset)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        shard_resolve_impl(summaries, shard_bb0, shard_bb_n, n_shards, symbol_clock, anchor_in, dibit_offset);
}
#endif

}  // namespace p25k
