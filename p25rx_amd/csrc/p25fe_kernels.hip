// p25fe_kernels.hip -- CDNA4 (gfx950) kernels of the P25 front end.
//
// Hot path of kchmck/p25rx re-designed for MI355X (stages of SURVEY.md section 8a):
//   K1 k_frontend : [u8 -> cf32] -> 5:1 decimating FIR -> channel FIR -> FM discriminator ->
//                   10-sample boxcar  == the five per-chunk loops of DemodTask::run
//                   (src/demod.rs:82-84, 87, 93, 109-111, 114) fused into one pass over HBM.
//   K2 k_sync     : frame-sync correlation + peak pick on the baseband (front half of
//                   MessageReceiver::feed, src/recv.rs:207) -> event flags + per-tile summary.
//   K3 k_scan     : per-channel scan of the tile summaries (symbol-timing anchor carry and
//                   dibit offsets) -- the serial state of the receiver turned into a scan.
//   K4 k_slice    : 4-level slicer at the anchored symbol instants -> dibits.
// Either side of the path (SURVEY.md section 8f / BASELINE.json config 3):
//   K0 k_predecim   : 2.4 Msps -> 240 ksps, 80-tap 10:1 decimating FIR (config 3's extra stage).
//   K5 k_nid        : network identifier after each frame sync, exhaustive BCH(63,16,23) search; k_chan_stats folds the
//                     records into the reference's per-channel statistics shape (src/hub.rs:557-581).
//   K6 k_channelise : one 2.4 Msps capture -> 192 channel streams at 240 ksps (factored 192-point DFT in registers).
// K0, K1, K2, K4, K6 (everything that touches sample data): one wave per workgroup, no s_barrier, short-lived workgroups --
// measured fastest for each of them.  K3 and K5 are small block-wide scans / searches.
//
// Arithmetic contract (docs/SPEC.md section 3): fp32, fma only where the spec says fma, single
// accumulator per output in tap order 0..T-1.  Compiled with -ffp-contract=off.
// Stencil / element-wise work: no MFMA.  HBM-bound by design: 8 B in + 0.8 B out per IQ sample.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "p25fe.h"
#include "p25fe_spec.h"

namespace p25k {

// ------------------------------------------------------------------------------------------
// geometry of K1.  A workgroup is ONE wave (64 lanes): no s_barrier anywhere, and the CU's resident
// waves (2-3 per SIMD) run their load / FIR / FM phases out of step with each other, which is what
// overlaps HBM latency, LDS traffic and VALU work.
// ------------------------------------------------------------------------------------------
constexpr int WV = 64;                       // lanes per workgroup
constexpr int DEC = P25FE_DECIM;
constexpr int T1 = P25FE_T1;
constexpr int T2 = P25FE_T2;
constexpr int BOX = P25FE_BOXCAR;
constexpr int HALO_Y = BOX;                  // y needed from m0-10 (fm needs y[m-1], boxcar needs fm[m-9])
constexpr int HALO_D = HALO_Y + (T2 - 1);    // 50: d needed from m0-50
constexpr int D_CARRY = T2 - 1;              // d carried between sub-tiles
// history (input samples before the first owned one) needed for exact results
constexpr int HIST_IQ = DEC * HALO_D + (T1 - 1) + (DEC - 1);   // 284

// PK = consecutive FIR outputs per lane (odd: lane stride 2*5*PK / 2*PK dwords -> conflict-free ds_read_b64)
template <int PK> struct Geo {
    static constexpr int P = PK;
    static constexpr int SUB = WV * PK;                   // decimated samples per sub-tile (320 for PK = 5)
    static constexpr int XWIN = DEC * SUB + (T1 - DEC);   // input samples feeding one sub-tile of d
    static constexpr int XIN_N = (XWIN + 1 + 7) / 8 * 8;       // window + the 0 / 1 sample alignment shift, rounded up
    static constexpr int D_N = D_CARRY + SUB;
    // LDS: [d carry 40][window region XIN_N][taps].  The sub-tile's new d samples and the output transpose OVERWRITE the
    // front of the window region once the decimator has consumed it (one wave: program order), so that 11 one-wave
    // workgroups fit a CU's 160 KB instead of 9 (occupancy is what bounds the overlap of HBM, LDS and VALU work).
    static constexpr size_t LDS_BYTES = sizeof(float2) * (D_CARRY + XIN_N) + sizeof(float) * (T1 + T2 + 3);
    static_assert(sizeof(float2) * XIN_N >= sizeof(float2) * SUB + sizeof(float) * SUB, "d and the transpose fit the window region");
    static constexpr int NBACK = (BOX - 1 + PK - 1) / PK; // lanes to the left whose fm values the boxcar needs
    static constexpr int WAVES_PER_SIMD = PK <= 3 ? 4 : 2;   // register budget the kernel is compiled for (128 / 256 VGPRs)
};

struct Taps {
    float dec[T1];
    float ch[T2];
};

// SPEC 3.4: polynomial atan2, identical operation sequence to the oracle's restatement.
__device__ __forceinline__ float spec_atan2f(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    const float t = mn / mx;
    const float s = t * t;
    float p = P25FE_ATAN_COEFFS[P25FE_ATAN_NCOEF - 1];
#pragma unroll
    for (int i = P25FE_ATAN_NCOEF - 2; i >= 0; --i) p = __builtin_fmaf(p, s, P25FE_ATAN_COEFFS[i]);
    float r = p * t;
    if (ay > ax) r = P25FE_HALF_PI - r;
    if (x < 0.0f) r = P25FE_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

// SPEC 3.4: FM discriminator, angle of s * conj(prev) times fs / (2 pi dev)   (src/demod.rs:54, 110)
__device__ __forceinline__ float fm_discriminate(float2 s, float2 prev)
{
    const float t = s.y * prev.y;
    const float re = __builtin_fmaf(s.x, prev.x, t);
    const float u = s.x * prev.y;
    const float im = __builtin_fmaf(s.y, prev.x, -u);
    return spec_atan2f(im, re) * P25FE_FM_GAIN;
}

// One complex sample from LDS as a single ds_read_b64.  The volatile 64-bit access keeps the compiler from
// pairing neighbouring reads into ds_read2_b64, which moves 16 B per lane at HALF the LDS rate of two
// ds_read_b64 (MI355X_MICROARCH.md LDS table: 8 vs 2+2 cycles per wave-instruction).
__device__ __forceinline__ float2 lds_read_c(const float2* p)
{
    typedef const volatile unsigned long long __attribute__((address_space(3))) * lds_u64_ptr;
    const unsigned long long u = *((lds_u64_ptr)p);                 // generic -> LDS address space: stays a DS op
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
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

// Phase boundary inside the one-wave workgroup: LDS operations of one wave execute in order, so only
// the compiler has to be kept from moving accesses across the boundary.
__device__ __forceinline__ void phase_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SPEC 3.1: rtlsdr_iq LUT value as arithmetic (src/demod.rs:82-84)
__device__ __forceinline__ float u8_to_f32(unsigned b) { return __builtin_fmaf((float)b, P25FE_U8_SCALE, -1.0f); }

// ------------------------------------------------------------------------------------------
// window loader: global -> registers, one 16-B vector per lane per load.
//
// Every load is unconditional and branch-free: the vector index is clamped into the range of
// vectors that contain at least one valid sample, and samples outside [-n_hist, n_new) are
// zeroed when the registers are written to LDS (only in boundary sub-tiles; interior ones skip
// the masking).  A clamped vector is 16-B aligned and shares its 16 bytes with a valid sample,
// so it can never touch an unmapped page.  (An earlier version had a second, element-wise path
// for straddling vectors; two paths writing the same VGPRs made the compiler put
// s_waitcnt vmcnt(0) in front of every load, serialising 13 HBM round trips per sub-tile.)
// ------------------------------------------------------------------------------------------
template <int FMT, int PK> struct Loader {
    using G = Geo<PK>;
    // A lane's vector always holds TWO samples -- 16 B of cf32 or 4 B of u8 pairs -- which become one 16-B LDS store,
    // lane-consecutive and therefore conflict-free.  (u8 used to load 16 B = 8 samples per lane: the 64-B lane stride of
    // the resulting ds_write_b128 is a 4-way bank conflict in every 8-lane group and cost 20 % of the kernel.)
    static constexpr int LOG_SPV = 1;
    static constexpr int SPV = 1 << LOG_SPV;
    static constexpr int NV = (G::XWIN + SPV + SPV * WV - 1) / (SPV * WV);   // vectors per lane: 13 for PK = 5
    using V = typename std::conditional<FMT == P25FE_FMT_CF32, uint4, unsigned>::type;
    V v[NV];

    // base: pointer to owned sample 0 of this channel; first: index of the first window sample;
    // i_last: last sample index this workgroup will ever need (loads past it collapse onto one cached vector)
    __device__ __forceinline__ void load(const void* base, long first, long n_hist, long n_new, long i_last, int tid)
    {
        const long v0 = first >> LOG_SPV;                           // floor division (arithmetic shift), uniform
        const V* q = reinterpret_cast<const V*>(base) + v0;         // uniform base of the window
        const long last = i_last < n_new - 1 ? i_last : n_new - 1;
        // clamp range relative to the window, saturated to 32 bits: one v_med3_i32 per vector
        long lo = ((-n_hist) >> LOG_SPV) - v0;                      // vector holding sample -n_hist
        long hi = (last >> LOG_SPV) - v0;                           // vector holding the last useful sample
        lo = lo < -(1L << 30) ? -(1L << 30) : (lo > (1L << 30) ? (1L << 30) : lo);
        hi = hi < -(1L << 30) ? -(1L << 30) : (hi > (1L << 30) ? (1L << 30) : hi);
        const int lo32 = (int)lo, hi32 = (int)hi;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int r = tid + j * WV;
            r = r < lo32 ? lo32 : r;
            r = r > hi32 ? hi32 : r;
#if defined(P25FE_ABLATE) && P25FE_ABLATE == 6     // measurement build: every window load hits one cached vector row
            r = tid;
            v[j] = reinterpret_cast<const V*>(base)[r];
            continue;
#endif
            v[j] = q[r];
        }
    }

    // XIN[k] receives sample first_al + k, first_al = first rounded down to a vector boundary: vector j of lane tid is
    // ONE aligned 16-B LDS store (ds_write_b128, lane-consecutive: conflict-free).  The consumers add the 0 / 1 sample
    // shift to their read base.  (Storing at XIN - shift made the alignment unknown at compile time: every staging
    // store became a ds_write2_b64 whose 16-B lane stride is a 2-way bank conflict -- 23 % of K1's LDS cycles.)
    __device__ __forceinline__ void store(float2* XIN, long first, long n_hist, long n_new, int tid) const
    {
        const long first_al = (first >> LOG_SPV) << LOG_SPV;
        const bool interior = first_al >= -n_hist && first_al + (long)NV * SPV * WV <= n_new;   // uniform
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
                    s[e] = make_float2(u8_to_f32(pair & 0xffu), u8_to_f32(pair >> 8));   // low byte = I (SPEC 3.1)
                }
                if (!interior) {
                    const long i = first_al + SPV * (long)(tid + j * WV) + e;
                    if (i < -n_hist || i >= n_new) s[e] = make_float2(0.f, 0.f);
                }
            }
            // only the last round of vectors can run past the window: everything else is stored unconditionally
            if (j < NV - 1 || SPV * (tid + j * WV) + SPV <= G::XIN_N) XV[tid + j * WV] = make_float4(s[0].x, s[0].y, s[1].x, s[1].y);
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
    long m_begin;           // first output to produce (<= 0: also outputs that lie in the history)
    float* power_partial;   // nullable: [n_channels][gridDim.x] partial sums of |y|^2
};

// CT = true: the handle's taps are the build's default tables (p25fe_spec.h) -> immediates, no
// registers or LDS reads spent on coefficients.  CT = false: caller-supplied taps, broadcast-read from LDS.
template <int FMT, bool CT, int PK>
__global__ __launch_bounds__(WV, (Geo<PK>::WAVES_PER_SIMD)) void k_frontend(K1Args a, const Taps* __restrict__ gtaps)
{
    using G = Geo<PK>;
    constexpr int SUB = G::SUB;
    constexpr int P = PK;
    constexpr int NBACK = G::NBACK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* D = reinterpret_cast<float2*>(smem);                    // [D_CARRY | SUB]: the SUB part aliases the window region
    float2* XIN = D + D_CARRY;                                      // 16-B aligned: staged with ds_write_b128
    float* OUT = reinterpret_cast<float*>(D + G::D_N);              // [SUB] output transpose, also inside the window region
    float* TAPS = reinterpret_cast<float*>(D + D_CARRY + G::XIN_N); // [T1 | T2], only when !CT
    const int tid = threadIdx.x;
    if (!CT) {
        for (int k = tid; k < T1; k += WV) TAPS[k] = gtaps->dec[k];
        for (int k = tid; k < T2; k += WV) TAPS[T1 + k] = gtaps->ch[k];
    }
    auto tap_dec = [&](int k) -> float { return CT ? P25FE_DEFAULT_DECIM_TAPS[k] : TAPS[k]; };
    auto tap_ch = [&](int k) -> float { return CT ? P25FE_DEFAULT_CHAN_TAPS[k] : TAPS[T1 + k]; };

    const long seg_len = (long)(SUB - HALO_D) + (long)(a.subs_per_seg - 1) * SUB;
    const long m_seg0 = a.m_begin + (long)blockIdx.x * seg_len;
    if (m_seg0 >= a.n_out) return;
    const long m_seg1 = (m_seg0 + seg_len < a.n_out) ? m_seg0 + seg_len : a.n_out;
    const int ch = blockIdx.y;
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)ch * a.ch_stride * (FMT == P25FE_FMT_CF32 ? 8 : 2);
    float* bb = a.bb + (size_t)ch * a.bb_stride;

    // zero the d carry (its garbage would only reach never-stored outputs, but keep it tidy)
    for (int k = tid; k < D_CARRY; k += WV) D[k] = make_float2(0.f, 0.f);

    Loader<FMT, PK> ld;
    long dlo = m_seg0 - HALO_D;                                    // first d index of this sub-tile
    const long i_last = (long)a.o0 + DEC * (m_seg1 - 1);          // newest input sample this segment needs
    ld.load(xb, (long)a.o0 + DEC * dlo - (T1 - 1), a.n_hist, a.n_new, i_last, tid);
    float pw = 0.f;

    // context carried across sub-tiles in wave-uniform registers: the last channel output and the
    // fm values of the last NBACK lanes of the previous sub-tile
    float2 y_carry = make_float2(0.f, 0.f);
    float f_carry[NBACK][P];
#pragma unroll
    for (int b = 0; b < NBACK; ++b)
#pragma unroll
        for (int p = 0; p < P; ++p) f_carry[b][p] = 0.f;

    // Outputs are kept in registers for one sub-tile and stored at the top of the next one, BEFORE the
    // prefetch loads are issued: vmcnt counts loads and stores in one in-order queue, so stores issued
    // after the prefetch would force the wait at the top of the loop to drain them too (measured: the
    // loop then ran at the HBM write round-trip per sub-tile).
    float outv[P];
#pragma unroll
    for (int q = 0; q < P; ++q) outv[q] = 0.f;
    int out_rel = -2 * SUB;                                         // out_lo - m_seg0; nothing in range yet
    const int seg_n = (int)(m_seg1 - m_seg0);
    float* const bb_seg = bb + m_seg0;
    auto flush_outputs = [&]() {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int r = out_rel + tid + q * WV;                   // output index relative to the segment start
            if (r >= 0 && r < seg_n) bb_seg[r] = outv[q];
        }
    };

    for (int it = 0; it < a.subs_per_seg; ++it, dlo += SUB) {
        if (dlo >= m_seg1) break;                                  // uniform
        const long first = (long)a.o0 + DEC * dlo - (T1 - 1);      // XIN[xsh + k] = x[first + k]
        const int xsh = (int)(first & 1);                           // window start relative to the aligned staging origin
        ld.store(XIN, first, a.n_hist, a.n_new, tid);
        phase_sync();
        flush_outputs();                                            // previous sub-tile's outputs (lane-predicated)
        // unconditional prefetch: past the segment's end the clamp makes every lane read one cached vector
        ld.load(xb, first + (long)DEC * SUB, a.n_hist, a.n_new, i_last, tid);

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 1      // measurement builds only (tools/ablate.sh): stop after the load pipeline
        continue;
#endif
        // ---- stage 2: 5:1 decimating FIR (src/demod.rs:87). Lane: d[dlo + P tid + p], p = 0..P-1.
        // Output p needs x[first + 5(P tid + p) + (T1-1) - k], k = 0..T1-1  -> XIN[5 P tid + 5p + 30 - k].
        {
            const float2* w = XIN + xsh + (DEC * P) * tid;
            float2 acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = DEC * (P - 1) + T1 - 1; j >= 0; --j) {    // newest to oldest => tap order 0..T1-1
                const float2 s = lds_read_c(w + j);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = DEC * p + (T1 - 1) - j;
                    if (k >= 0 && k < T1) {
                        acc[p].x = __builtin_fmaf(tap_dec(k), s.x, acc[p].x);
                        acc[p].y = __builtin_fmaf(tap_dec(k), s.y, acc[p].y);
                    }
                }
            }
            // d overwrites the front of the window: every lane's window reads must be complete first.  (The compiler
            // reasons per thread and could prove a lane's own store and loads disjoint -- the fence orders the wave.)
            phase_sync();
#pragma unroll
            for (int p = 0; p < P; ++p) D[D_CARRY + P * tid + p] = acc[p];
        }
        phase_sync();

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 2
        continue;
#endif
        // ---- stage 3: channel FIR (src/demod.rs:93). Lane: y[dlo + P tid + p] from D[P tid + p + 40 - k].
        float2 y[P];
        {
            const float2* w = D + P * tid;
#pragma unroll
            for (int p = 0; p < P; ++p) y[p] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = (P - 1) + T2 - 1; j >= 0; --j) {
                const float2 s = lds_read_c(w + j);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = p + (T2 - 1) - j;
                    if (k >= 0 && k < T2) {
                        y[p].x = __builtin_fmaf(tap_ch(k), s.x, y[p].x);
                        y[p].y = __builtin_fmaf(tap_ch(k), s.y, y[p].y);
                    }
                }
            }
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
        phase_sync();                                               // all reads of D done before its carry is rewritten
        // carry the d history to the next sub-tile (ordered before its next use by the next phase boundaries)
        for (int k = tid; k < D_CARRY; k += WV) D[k] = D[SUB + k];

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 3
        continue;
#endif
        // ---- stage 4: FM discriminator (src/demod.rs:109-111) on the lane's own P outputs; the sample before
        // the first one comes from lane-1 (DPP), for lane 0 from the previous sub-tile (SGPR carry)
        float f[P];
        {
            float2 yprev;
            yprev.x = wave_shr1(y[P - 1].x, y_carry.x);
            yprev.y = wave_shr1(y[P - 1].y, y_carry.y);
            f[0] = fm_discriminate(y[0], yprev);
#pragma unroll
            for (int p = 1; p < P; ++p) f[p] = fm_discriminate(y[p], y[p - 1]);
            y_carry.x = lane_bcast<WV - 1>(y[P - 1].x);
            y_carry.y = lane_bcast<WV - 1>(y[P - 1].y);
        }

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 4
        asm volatile("" ::"v"(f[0]), "v"(f[P - 1]));
        continue;
#endif
        // ---- stage 5: boxcar (src/demod.rs:114): b[m] = (fm[m] + fm[m-1] + ... + fm[m-9]) / 10, newest first.
        // fm[P tid + p - j] lives in this lane (q = p - j >= 0) or in lane - b, b = ceil(-q / P): prevf[b-1][q + b P].
        {
            float prevf[NBACK][P];
#pragma unroll
            for (int p = 0; p < P; ++p) prevf[0][p] = wave_shr1(f[p], f_carry[0][p]);
#pragma unroll
            for (int b = 1; b < NBACK; ++b)
#pragma unroll
                for (int p = 0; p < P; ++p) prevf[b][p] = wave_shr1(prevf[b - 1][p], f_carry[b][p]);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float acc = f[p];
#pragma unroll
                for (int j = 1; j < BOX; ++j) {
                    const int q = p - j;
                    if (q >= 0) {
                        acc = acc + f[q];
                    } else {
                        const int b = (-q + P - 1) / P;             // 1 .. NBACK
                        acc = acc + prevf[b - 1][q + b * P];
                    }
                }
                OUT[P * tid + p] = acc * P25FE_BOXCAR_SCALE;        // lane stride P dwords (odd): conflict-free
            }
            // next sub-tile's lane 0 / 1 / ... read these: lane 63 is one lane back, lane 62 two, ...
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (NBACK >= 3) f_carry[2][p] = lane_bcast<WV - 3>(f[p]);
                if (NBACK >= 2) f_carry[1][p] = lane_bcast<WV - 2>(f[p]);
                f_carry[0][p] = lane_bcast<WV - 1>(f[p]);
            }
        }
        phase_sync();
        // transpose through LDS: lane-consecutive outputs -> coalesced (deferred) global stores
#pragma unroll
        for (int q = 0; q < P; ++q) outv[q] = OUT[tid + q * WV];
        out_rel = (int)(dlo - m_seg0);
        phase_sync();
    }
    flush_outputs();

    if (a.power_partial) {
        // wave reduction of the |y|^2 partials (tree; tolerance vs the sequential fold is in the tests)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) pw = pw + __shfl_down(pw, d, 64);
        if (tid == 0) a.power_partial[(size_t)ch * gridDim.x + blockIdx.x] = pw;
    }
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
constexpr int K0_P = 3;                      // outputs per lane (odd)
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
constexpr int K0_JP = 205;                   // entries per phase row (>= K0_NIN / PD + 2); = 13 mod 16: staging position t lands on
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
#ifndef P25FE_CZ_G
#define P25FE_CZ_G 4
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
__constant__ CzTw CZ_TW = cz_make_tw();
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

    // c1 in groups of CZ_G (a rolled loop: the c1-dependent twiddles come from CZ_TW by scalar loads): the 80 products
    // a[p] = h[p] x[n - p] are re-formed from LDS once per group (window position 79 + 10 lane - p -> phase
    // 9 - p % 10, column lane + 7 - p / 10) instead of living in 160 registers
#if defined(P25FE_ABLATE6)                          // measurement build (tools/k6_variants.sh): the store stream alone
    {
        const float2 xv = lds_read_c(X + lane);
#pragma unroll 1
        for (int c = 0; c < CZ_M; ++c) yb[(size_t)c * a.y_stride] = make_float2(xv.x + (float)c, xv.y);
        return;
    }
#endif
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
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float2 rot = ROT[ix];
                const float2 yv = BB[q][4 * e + f];
                const float2 o = make_float2(__builtin_fmaf(-yv.y, rot.y, yv.x * rot.x), __builtin_fmaf(yv.y, rot.x, yv.x * rot.y));
                {   // streaming store (nt): 22 GB of output per minute of capture never fit a cache; measured +4.5 %.
                    // Rows are padded to whole tiles: no predicate.
                    typedef float __attribute__((ext_vector_type(2))) f32x2;
                    f32x2 ov; ov.x = o.x; ov.y = o.y;
                    __builtin_nontemporal_store(ov, reinterpret_cast<f32x2*>(yb + (size_t)(c1 + CZ_C1 * (e + 4 * f)) * a.y_stride));
                }
                ix += step8; ix = ix >= (unsigned)CZ_M ? ix - CZ_M : ix;
            }
        }
        idx1 += nm; idx1 = idx1 >= (unsigned)CZ_M ? idx1 - CZ_M : idx1;
        }
    }
}

// finish power_dbm (src/demod.rs:123-134): 30 + 10 log10( (sum / N) / R ), R = 1
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
        out_dbm[ch] = n > 0 ? 30.0f + 10.0f * log10f(avg / 1.0f) : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// K2..K4: symbol receiver as a scan (SPEC 3.7-3.8)
//
// A detection at s (sync word's last symbol) is DECIDED at e = s + W (the peak window is complete)
// and governs instants n > e: instant n is
// governed by the detection with the latest e <= n.  Tiles own the EVENTS whose e falls in
// them, so every dependency points left.
// ------------------------------------------------------------------------------------------
constexpr int W = P25FE_PEAK_W;
constexpr int SPS = P25FE_SPS;
constexpr int SYNC_SPAN = P25FE_SYNC_SPAN;                   // 230
constexpr int TB = 1024;                                     // baseband samples per tile
constexpr int HIST_BB = SYNC_SPAN + 2 * W;                   // 240: left context of a tile

struct TileRec {            // per (channel, tile) summary written by K2
    long first_event;       // absolute decision index e = s + W of the tile's first event, -1 if none
    long last_s;            // s of the tile's last event (absolute), valid if first_event >= 0
    float hi, mid, lo;      // thresholds of the last event
    int n_events;
    long post_count;        // instants in (first_event, tile_end) under the tile's own events
};

// Packed per-tile summary for the scan (one coalesced 8-byte word per tile):
//   bits  0..12  first_off + 1   (0: the tile has no event)      bits 13..25  last_off + 1
//   bits 26..38  n_events                                          bits 39..51  post_count
constexpr int TS_BITS = 13;                                  // TB + 1 <= 2^13
constexpr unsigned long long TS_MASK = (1ull << TS_BITS) - 1;
__host__ __device__ inline unsigned long long pack_tsum(int first_off, int last_off, int n_events, int post_count)
{
    return (unsigned long long)(first_off + 1) | ((unsigned long long)(last_off + 1) << TS_BITS) |
           ((unsigned long long)n_events << (2 * TS_BITS)) | ((unsigned long long)post_count << (3 * TS_BITS));
}

struct ScanOut {            // per (channel, tile) carry-in written by K3
    int src;                        // tile whose last event is this tile's carry-in anchor; -1: the range's anchor_in
    unsigned event_off;             // events of the range before this tile
    unsigned long long dibit_off;   // dibits of the range before this tile
};

__device__ __forceinline__ float bb_at(const float* bbp, long n_hist, long n, long i)
{
    return (i >= -n_hist && i < n) ? bbp[i] : 0.0f;
}

// thresholds from the sync word ending at local LDS index `is` (SPEC 3.8)
__device__ __forceinline__ void sync_thresholds(const float* BT, int is, float& hi, float& mid, float& lo)
{
    float Pp = 0.f, Nn = 0.f;
#pragma unroll
    for (int j = 0; j < P25FE_SYNC_DIBITS; ++j) {
        const float v = BT[is - SPS * (P25FE_SYNC_DIBITS - 1 - j)];
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

// the same from global memory (s = range-local index of the sync word's last symbol): same operations, same bits
__device__ __forceinline__ void sync_thresholds_global(const float* bbp, long n_hist, long n, long s, float& hi, float& mid, float& lo)
{
    float Pp = 0.f, Nn = 0.f;
#pragma unroll
    for (int j = 0; j < P25FE_SYNC_DIBITS; ++j) {
        const float v = bb_at(bbp, n_hist, n, s - SPS * (P25FE_SYNC_DIBITS - 1 - j));
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

// number of n in [lo, hi) with n > s and (n - s) % SPS == 0   (closed form)
__host__ __device__ inline long count_instants(long s, long lo, long hi)
{
    if (lo <= s) lo = s + 1;
    if (hi <= lo) return 0;
    // first k with s + SPS*k >= lo
    const long k0 = (lo - s + SPS - 1) / SPS;
    const long k1 = (hi - 1 - s) / SPS;           // last k with s + SPS*k <= hi-1
    return k1 >= k0 ? k1 - k0 + 1 : 0;
}

// wave-level inclusive max scan (64 lanes)
__device__ __forceinline__ long wave_incl_max(long v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long o = __shfl_up(v, d, 64);
        if (lane >= d) v = o > v ? o : v;
    }
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

struct SyncArgs {
    const float* bb;        // owned baseband sample 0 of channel 0
    long bb_stride;
    long n_hist;            // valid samples before it
    long n;                 // owned samples per channel
    long abs0;              // absolute index of owned sample 0
    int n_tiles;
    uint8_t* events;        // [ch][n] event flags (e-indexed)
    long ev_stride;
    TileRec* recs;          // [ch][n_tiles]
    unsigned long long* tsum;   // [ch][n_tiles] packed summaries for K3
};

// K2: correlate, peak-pick, flag events, summarise the tile.  Built like K1: one wave per workgroup (no s_barrier,
// wave-level scans only), 16-B window loads.  K2_SUBS > 1 walks consecutive tiles with the next window prefetched into
// registers; measured on config 2: 1 tile per workgroup 0.070 ms, 2: 0.077, 4: 0.081, 8: 0.089 -- short-lived waves win.
//   correlation: lanes 0..59 = 10 sample phases x 6 slot blocks; a lane produces K2_CQ = 18 symbol-spaced outputs
//   from a sliding window of 41 LDS reads (2.3 reads per output instead of 24), each output with its own
//   accumulators in tap order j = 0..23 (SPEC 3.7);
//   peak pick / instant count: a lane owns 16 consecutive samples (<= 2 symbol instants without an event inside).
#ifndef P25FE_K2_SUBS
#define P25FE_K2_SUBS 1
#endif
constexpr int K2_SUBS = P25FE_K2_SUBS;
constexpr int K2_CQ = 18;
constexpr int K2_LANES = SPS * 6;                                // 60 lanes busy in the correlation
constexpr int K2_NC = TB + 2 * W;                                // c[] positions a tile needs: 1034
constexpr int K2_VPL = TB / WV;                                  // 16 samples per lane
constexpr int K2_NV = 6;                                         // 16-B vectors per lane: window of 1310 (+3 shift) floats
constexpr int K2_BT = 4 * WV * K2_NV;                            // 1536 floats of LDS (every vector has a slot)
static_assert(SPS * K2_CQ * (K2_LANES / SPS) >= K2_NC, "correlation lanes must cover the tile");
static_assert(SPS * K2_CQ * (K2_LANES / SPS - 1) + SPS - 1 + SPS * (K2_CQ + P25FE_SYNC_DIBITS - 2) + 3 < K2_BT, "window fits");
static_assert(K2_VPL > SPS && K2_VPL <= 2 * SPS, "instant count closed form assumes 1..2 instants per lane");

__device__ __forceinline__ int wave_incl_max_i(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(WV, 3) void k_sync(SyncArgs a)
{
    __shared__ __attribute__((aligned(16))) float BT[K2_BT];      // BT[k] = b[t0 - HIST_BB + k]
    float* const CT = BT;                                         // CT[k] = c[t0 - 2W + k] OVERWRITES the window once every lane
                                                                  // has its taps in registers (one wave: program order) --
                                                                  // 7.2 KB of LDS per wave instead of 11.3: 22 waves per CU
    __shared__ __attribute__((aligned(8))) uint8_t CAND[K2_NC + 16];   // flag of position k at CAND[k - W + 8]: sample i's flag at the 8-aligned CAND[i + 8]

    const int lane = threadIdx.x, ch = blockIdx.y;
    const float* bbp = a.bb + (size_t)ch * a.bb_stride;
    const long A = (long)(reinterpret_cast<uintptr_t>(bbp) >> 2);          // float address of owned sample 0
    const float4* vec0 = reinterpret_cast<const float4*>(reinterpret_cast<uintptr_t>(bbp) & ~(uintptr_t)15);
    const long Av = A >> 2;                                                // vector address of vec0 (the vector holding sample 0)
    const long vlo = ((A - a.n_hist) >> 2) - Av, vhi = ((A + a.n - 1) >> 2) - Av;   // loadable vectors (relative)

    float4 v[K2_NV];
    auto load = [&](int tile) {
        const long P = A + (long)tile * TB - HIST_BB;                      // float address of BT[0]
        const long V0 = (P >> 2) - Av;
        long lo = vlo - V0, hi = vhi - V0;
        lo = lo < -(1L << 30) ? -(1L << 30) : (lo > (1L << 30) ? (1L << 30) : lo);
        hi = hi < -(1L << 30) ? -(1L << 30) : (hi > (1L << 30) ? (1L << 30) : hi);
        const int lo32 = (int)lo, hi32 = (int)hi;
        const float4* q = vec0 + V0;
#pragma unroll
        for (int j = 0; j < K2_NV; ++j) {
            int r = lane + j * WV;
            r = r < lo32 ? lo32 : r;
            r = r > hi32 ? hi32 : r;
            v[j] = q[r];
        }
    };
    auto stage = [&](int tile) {
        const long t0 = (long)tile * TB;
        const long P = A + t0 - HIST_BB;
        const int sh = (int)(P & 3);
        const bool fast = sh == 0 && t0 - HIST_BB >= -a.n_hist && t0 - HIST_BB + K2_BT <= a.n;   // uniform
        if (fast) {
#pragma unroll
            for (int j = 0; j < K2_NV; ++j) *reinterpret_cast<float4*>(&BT[4 * (lane + j * WV)]) = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < K2_NV; ++j) {
                const float w4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * (lane + j * WV) + e - sh;
                    const long i = t0 - HIST_BB + k;
                    if (k >= 0 && k < K2_BT) BT[k] = (i >= -a.n_hist && i < a.n) ? w4[e] : 0.0f;
                }
            }
        }
    };

    const int tile_first = blockIdx.x * K2_SUBS;
    load(tile_first);
#pragma unroll 1
    for (int it = 0; it < K2_SUBS; ++it) {
        const int tile = tile_first + it;
        if (tile >= a.n_tiles) break;                               // uniform
        const long t0 = (long)tile * TB;                            // local index of the tile's first sample
        const int tn = (a.n - t0 < TB) ? (int)(a.n - t0) : TB;      // samples in this tile
        stage(tile);
        phase_sync();
        if (K2_SUBS > 1 && it + 1 < K2_SUBS && tile + 1 < a.n_tiles) load(tile + 1);   // uniform; prefetch the next window
#if defined(P25FE_ABLATE2) && P25FE_ABLATE2 <= 1   // measurement builds only: stop after the LDS staging
        if (lane == 0) a.tsum[(size_t)ch * a.n_tiles + tile] = BT[tile % K2_BT] == 123.f ? 1ull : 0ull;
        phase_sync();
        continue;
#endif

        // c[s], cand[s] for s = t0 - 2W + k, k in [0, TB + 2W): needs b[s - 230 .. s] = BT[k + 10 j], j = 0..23.
        if (lane < K2_LANES) {
            const int r = lane % SPS, qb = lane / SPS;
            const int base = SPS * K2_CQ * qb + r;                  // k of the lane's first output
            float w[K2_CQ + P25FE_SYNC_DIBITS - 1];
#pragma unroll
            for (int m = 0; m < K2_CQ + P25FE_SYNC_DIBITS - 1; ++m) w[m] = BT[base + SPS * m];
#pragma unroll
            for (int i = 0; i < K2_CQ; ++i) {
                const int k = base + SPS * i;
                float c = 0.f, e = 0.f;
#pragma unroll
                for (int j = 0; j < P25FE_SYNC_DIBITS; ++j) {
                    const float x = w[i + j];
                    c = ((P25FE_SYNC_SIGN_MASK >> j) & 1u) ? c + x : c - x;
                    e = __builtin_fmaf(x, x, e);
                }
                if (k < K2_NC) {
                    CT[k] = c;
                    CAND[k - W + 8] = (c > 0.0f) && (e >= P25FE_SYNC_E_MIN) && (c * c >= P25FE_SYNC_RHO2_N * e);
                }
            }
        }
        phase_sync();
#if defined(P25FE_ABLATE2) && P25FE_ABLATE2 <= 2   // stop after the correlation
        if (lane == 0) a.tsum[(size_t)ch * a.n_tiles + tile] = (CT[tile % K2_NC] == 123.f || CAND[tile % K2_NC]) ? 1ull : 0ull;
        phase_sync();
        continue;
#endif

        // event at local index i (decided when sample t0 + i arrives) <=> detection at s = t0 + i - W -> CT index
        // k = i + W.  The peak window s + W = t0 + i - 1 < n is guaranteed by i < tn.  CAND is stored shifted so that
        // a lane's 16 flags are two aligned 8-byte LDS reads; no candidate among them (the common case) skips the test.
        int my_last = -1, my_first = -1, my_ev = 0;
        unsigned evw[4] = {0u, 0u, 0u, 0u};                         // 16 event flags, one byte each
        {
            const uint2 c0 = *reinterpret_cast<const uint2*>(&CAND[lane * K2_VPL + 8]);
            const uint2 c1 = *reinterpret_cast<const uint2*>(&CAND[lane * K2_VPL + 16]);
            const unsigned cw[4] = {c0.x, c0.y, c1.x, c1.y};
            if ((c0.x | c0.y | c1.x | c1.y) != 0u) {
#pragma unroll
                for (int u = 0; u < K2_VPL; ++u) {
                    const int i = lane * K2_VPL + u;
                    const int k = i + W;
                    const bool cand = ((cw[u >> 2] >> (8 * (u & 3))) & 0xffu) != 0u;
                    if (i < tn && cand) {
                        const float cm = CT[k];
                        bool det = true;
#pragma unroll
                        for (int d = 1; d <= W; ++d) det = det && (cm > CT[k - d]) && (cm >= CT[k + d]);
                        if (det) {
                            evw[u >> 2] |= 1u << (8 * (u & 3));
                            if (my_first < 0) my_first = i;
                            my_last = i; ++my_ev;
                        }
                    }
                }
            }
        }
        if (a.events) {
            uint8_t* evp = a.events + (size_t)ch * a.ev_stride + t0;
            if (lane * K2_VPL + K2_VPL <= tn && ((a.ev_stride & 15) == 0)) {
                *reinterpret_cast<uint4*>(evp + lane * K2_VPL) = make_uint4(evw[0], evw[1], evw[2], evw[3]);
            } else {
#pragma unroll
                for (int u = 0; u < K2_VPL; ++u)
                    if (lane * K2_VPL + u < tn) evp[lane * K2_VPL + u] = (uint8_t)((evw[u >> 2] >> (8 * (u & 3))) & 1u);
            }
        }
#if defined(P25FE_ABLATE2) && P25FE_ABLATE2 <= 3   // stop after peak pick + event flags
        if (my_last == 12345) a.tsum[(size_t)ch * a.n_tiles + tile] = 1ull;
        phase_sync();
        continue;
#endif

        // latest own event before each lane's first sample, then count instants under own events
        const int incl = wave_incl_max_i(my_last, lane);
        int incoming = __shfl_up(incl, 1, 64);
        if (lane == 0) incoming = -1;
        int cur = incoming;                                         // tile-local index of the governing event, -1: carry-in (unknown here)
        int cnt = 0;
        const int i0 = lane * K2_VPL;
        if (my_ev == 0) {
            // common case: no event inside my 16 samples -> one or two instants, closed form
            if (cur >= 0) {
                const int navail = tn - i0 < K2_VPL ? tn - i0 : K2_VPL;
                const unsigned ph = (unsigned)(i0 - (cur - W)) % (unsigned)SPS;
                const int f = (int)((SPS - ph) % (unsigned)SPS);
                cnt = (f < navail ? 1 : 0) + (f + SPS < navail ? 1 : 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < K2_VPL; ++u) {
                const int i = i0 + u;
                if (i < tn) {
                    if (cur >= 0) {                                 // events decided BEFORE this sample govern it
                        const unsigned dist = (unsigned)(i - (cur - W));   // > 0: distance to the anchor
                        if (dist % (unsigned)SPS == 0u) ++cnt;
                    }
                    if ((evw[u >> 2] >> (8 * (u & 3))) & 1u) cur = i;
                }
            }
        }
        const int total_packed = wave_sum_i(cnt | (my_ev << 16));   // instants low half, events high half
        const int total_cnt = total_packed & 0xffff, total_ev = total_packed >> 16;
        const int last_ev = __shfl(incl, 63, 64);
        const unsigned long long evmask = __ballot(my_ev > 0);
        const int first_ev = evmask ? __shfl(my_first, __builtin_ctzll(evmask), 64) : -1;
        if (lane == 0) {
            TileRec r;
            r.first_event = first_ev >= 0 ? a.abs0 + t0 + first_ev : -1;
            r.n_events = total_ev;
            r.post_count = total_cnt;
            r.last_s = -1; r.hi = r.mid = r.lo = 0.f;
            if (last_ev >= 0) {
                const int s = last_ev - W;                          // tile-local
                float hi, mid, lo;
                sync_thresholds_global(bbp, a.n_hist, a.n, t0 + s, hi, mid, lo);   // rare; the LDS window is gone
                r.last_s = a.abs0 + t0 + s; r.hi = hi; r.mid = mid; r.lo = lo;
            }
            a.recs[(size_t)ch * a.n_tiles + tile] = r;
            a.tsum[(size_t)ch * a.n_tiles + tile] =
                first_ev >= 0 ? pack_tsum(first_ev, last_ev, total_ev, total_cnt) : 0ull;
        }
        phase_sync();                                               // BT / CT / CAND are rewritten by the next tile
    }
}

// The scan over tiles is two-level so that it does not run on a single CU: K3a scans groups of K3_GROUP tiles in
// parallel (one workgroup each, no carry-in), K3b walks the (few) group aggregates and hands every group its carry,
// K4 adds the group carry to the group-local values.
struct GroupAgg {           // per (channel, group), written by K3a
    int first_tile;         // absolute index of the group's first tile with an event, -1 if none
    int last_tile;          // ... last tile with an event
    long first_event;       // absolute decision index of the group's first event
    unsigned long long cnt_after_first;   // dibits governed by the group's own events
    unsigned long long n_events;
};
struct GroupCarry {         // per (channel, group), written by K3b
    long anchor_s;          // anchor in force at the group's first sample
    float hi, mid, lo;
    int valid;
    unsigned long long dibit_base;         // dibits of the range before the group
    unsigned long long base_after_first;   // dibit_base + dibits before the group's first own event
    unsigned long long event_base;         // events of the range before the group
};

struct ScanArgs {
    const TileRec* recs;
    const unsigned long long* tsum;
    ScanOut* outs;          // group-local: src absolute or -1, offsets relative to the group's first own event / start
    GroupAgg* aggs;         // [ch][n_groups]
    GroupCarry* carries;    // [ch][n_groups]
    int n_tiles;
    int n_groups;
    int n_channels;
    long n;                 // owned samples per channel
    long abs0;
    const p25fe_anchor_t* anchor_in;    // nullable, [ch]
    p25fe_result_t* result;             // [ch]
    unsigned long long n_baseband;      // to report
};

constexpr int NT3 = 1024;

// 1024-thread inclusive scans: wave shuffles, then the 16 wave totals through LDS.
__device__ __forceinline__ long block_incl_max1024(long v, long* sh, int tid, long& total)
{
    const int lane = tid & 63, wv = tid >> 6;
    long inc = wave_incl_max(v, lane);
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    long carry = -1, tot = -1;
#pragma unroll
    for (int k = 0; k < NT3 / 64; ++k) {
        const long t = sh[k];
        if (k < wv) carry = t > carry ? t : carry;
        tot = t > tot ? t : tot;
    }
    __syncthreads();
    total = tot;
    return inc > carry ? inc : carry;
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
__device__ __forceinline__ unsigned long long block_incl_sum1024(unsigned long long v, unsigned long long* sh, int tid,
                                                                  unsigned long long& total)
{
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long inc = wave_incl_sum64(v, lane);
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    unsigned long long carry = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT3 / 64; ++k) {
        const unsigned long long t = sh[k];
        if (k < wv) carry += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return inc + carry;
}

constexpr int K3_GROUP = 2048;                                   // tiles per K3a workgroup (2 per thread, 16 KB of LDS)

// K3a: one workgroup per (group of K3_GROUP tiles, channel), no carry-in.  The packed tile summaries are staged in
// LDS; every thread walks a contiguous run of tiles; two block scans (latest event tile; dibit / event counts).
__global__ __launch_bounds__(NT3) void k_scan(ScanArgs a)
{
    __shared__ unsigned long long TS[K3_GROUP];
    __shared__ long shl[NT3 / 64];
    __shared__ unsigned long long shu[NT3 / 64];
    __shared__ long excl_tmp[NT3 / 64];
    const int tid = threadIdx.x, grp = blockIdx.x, ch = blockIdx.y;
    const unsigned long long* tsum = a.tsum + (size_t)ch * a.n_tiles;
    ScanOut* outs = a.outs + (size_t)ch * a.n_tiles;
    const int c0 = grp * K3_GROUP;
    const int cn = (a.n_tiles - c0 < K3_GROUP) ? a.n_tiles - c0 : K3_GROUP;

    for (int k = tid; k < cn; k += NT3) TS[k] = tsum[c0 + k];
    __syncthreads();
    const int per = (cn + NT3 - 1) / NT3;
    const int k0 = tid * per, k1 = (k0 + per < cn) ? k0 + per : cn;

    // pass 1: latest event tile inside my run -> block exclusive max
    long last = -1;
    for (int k = k0; k < k1; ++k) if (TS[k] & TS_MASK) last = k;
    long tot_max;
    const long incl = block_incl_max1024(last, shl, tid, tot_max);
    long excl = __shfl_up(incl, 1, 64);
    if ((tid & 63) == 63) excl_tmp[tid >> 6] = incl;
    __syncthreads();
    if ((tid & 63) == 0) excl = tid > 0 ? excl_tmp[(tid >> 6) - 1] : -1L;
    __syncthreads();

    // passes 2 and 3 walk the run with the anchor expressed as a PHASE: ph = (tile_start - s) mod 10, so the
    // closed-form instant counts are 32-bit.  Tiles before the group's first event (src < 0) are governed by the
    // group's carry-in, which is not known here: they contribute nothing and K4 adds their part from GroupCarry.
    auto phase_at = [&](int k, long src, bool& v) -> unsigned {   // phase of tile k's start under anchor `src`
        v = src >= 0;
        if (!v) return 0u;
        const int last_off = (int)((TS[src] >> TS_BITS) & TS_MASK) - 1;
        const unsigned dist = (unsigned)(k - (int)src) * (unsigned)TB - (unsigned)last_off + (unsigned)W;   // tile_start - s > 0
        return dist % (unsigned)SPS;
    };
    auto count32 = [&](unsigned ph, int len) -> unsigned {        // n in [0, len): (ph + n) % 10 == 0
        const int f = (int)((SPS - ph) % (unsigned)SPS);          // first instant offset
        return len > f ? (unsigned)(len - f + SPS - 1) / (unsigned)SPS : 0u;
    };
    auto tile_len = [&](int k) -> int {
        const long rem = a.n - (long)(c0 + k) * TB;
        return rem < TB ? (int)rem : TB;
    };
    unsigned long long my_cnt = 0, my_ev = 0;
    {
        long src = excl;
        bool v = false;
        unsigned ph = k0 < k1 ? phase_at(k0, src, v) : 0u;
        for (int k = k0; k < k1; ++k) {
            const unsigned long long u = TS[k];
            const int first1 = (int)(u & TS_MASK);
            const int len = first1 ? first1 : tile_len(k);        // instant AT the event index is still the old anchor's
            my_cnt += (v ? count32(ph, len) : 0u) + (unsigned)((u >> (3 * TS_BITS)) & TS_MASK);
            my_ev += (u >> (2 * TS_BITS)) & TS_MASK;
            if (first1) {
                src = k; v = true;
                const int last_off = (int)((u >> TS_BITS) & TS_MASK) - 1;
                ph = (unsigned)(TB - last_off + W) % (unsigned)SPS;     // next tile's start under the new anchor
            } else {
                ph = (ph + (unsigned)TB) % (unsigned)SPS;
            }
        }
    }
    unsigned long long tot_cnt, tot_ev;
    const unsigned long long icnt = block_incl_sum1024(my_cnt, shu, tid, tot_cnt);
    const unsigned long long iev = block_incl_sum1024(my_ev, shu, tid, tot_ev);

    // pass 3: write the per-tile group-local carry-ins
    {
        long src = excl;
        bool v = false;
        unsigned ph = k0 < k1 ? phase_at(k0, src, v) : 0u;
        unsigned long long dc = icnt - my_cnt, ec = iev - my_ev;
        for (int k = k0; k < k1; ++k) {
            const unsigned long long u = TS[k];
            const int first1 = (int)(u & TS_MASK);
            const int len = first1 ? first1 : tile_len(k);
            const unsigned pre = v ? count32(ph, len) : 0u;
            ScanOut o;
            o.src = src >= 0 ? (int)(c0 + src) : -1;
            o.event_off = (unsigned)ec;
            o.dibit_off = dc;
            outs[c0 + k] = o;
            dc += pre + (unsigned)((u >> (3 * TS_BITS)) & TS_MASK);
            ec += (u >> (2 * TS_BITS)) & TS_MASK;
            if (first1) {
                src = k; v = true;
                const int last_off = (int)((u >> TS_BITS) & TS_MASK) - 1;
                ph = (unsigned)(TB - last_off + W) % (unsigned)SPS;
            } else {
                ph = (ph + (unsigned)TB) % (unsigned)SPS;
            }
        }
    }
    // group aggregate: first / last event tile (min over first-event holders = the thread with excl < 0 and an event)
    __shared__ int first_tile_sh;
    if (tid == 0) first_tile_sh = -1;
    __syncthreads();
    if (last >= 0 && excl < 0) {
        int f = -1;
        for (int k = k1 - 1; k >= k0; --k) if (TS[k] & TS_MASK) f = k;
        first_tile_sh = f;
    }
    __syncthreads();
    if (tid == 0) {
        GroupAgg g;
        g.first_tile = first_tile_sh >= 0 ? c0 + first_tile_sh : -1;
        g.last_tile = tot_max >= 0 ? c0 + (int)tot_max : -1;
        g.first_event = -1;
        if (first_tile_sh >= 0)
            g.first_event = a.abs0 + (long)(c0 + first_tile_sh) * TB + (long)(TS[first_tile_sh] & TS_MASK) - 1;
        g.cnt_after_first = tot_cnt;
        g.n_events = tot_ev;
        a.aggs[(size_t)ch * a.n_groups + grp] = g;
    }
}

// K3b: one thread per channel walks the group aggregates (a 1-hour capture has ~80 groups) and hands every group its
// carry-in anchor and bases; also writes the range summary.
// K3b: one WAVE per channel.  The walk over the group aggregates is serial (each group's carry-in is the previous one's
// carry-out), but its inputs are not: lane g fetches group g's aggregate and the TileRec of its last event tile (two
// dependent loads, all lanes at once) into LDS, then lane 0 walks LDS.  A one-thread walk with the loads inside took 9 us
// for the 14 groups of config 2.
__global__ __launch_bounds__(WV) void k_scan_groups(ScanArgs a)
{
    __shared__ GroupAgg AG[WV];
    __shared__ TileRec TR[WV];
    const int ch = blockIdx.x, lane = threadIdx.x;
    const TileRec* recs = a.recs + (size_t)ch * a.n_tiles;
    p25fe_anchor_t A;
    A.valid = 0; A.s = 0; A.hi = A.mid = A.lo = 0.f;
    if (a.anchor_in) A = a.anchor_in[ch];
    unsigned long long B = 0, E = 0, base_first = 0;
    long first_event = -1;
    for (int g0 = 0; g0 < a.n_groups; g0 += WV) {
        const int ng = a.n_groups - g0 < WV ? a.n_groups - g0 : WV;
        if (lane < ng) {
            const GroupAgg ag = a.aggs[(size_t)ch * a.n_groups + g0 + lane];
            AG[lane] = ag;
            if (ag.first_tile >= 0) TR[lane] = recs[ag.last_tile];
        }
        phase_sync();
        if (lane == 0) {
            for (int k = 0; k < ng; ++k) {
                const int g = g0 + k;
                const GroupAgg ag = AG[k];
                const long lo = a.abs0 + (long)g * K3_GROUP * TB;
                long hi = lo + (long)K3_GROUP * TB;
                if (hi > a.abs0 + a.n) hi = a.abs0 + a.n;
                GroupCarry c;
                c.anchor_s = A.s; c.hi = A.hi; c.mid = A.mid; c.lo = A.lo; c.valid = A.valid;
                c.dibit_base = B;
                c.event_base = E;
                const long pre_hi = ag.first_tile >= 0 ? ag.first_event + 1 : hi;
                const unsigned long long lead = A.valid ? (unsigned long long)count_instants(A.s, lo, pre_hi) : 0ull;
                c.base_after_first = B + lead;
                a.carries[(size_t)ch * a.n_groups + g] = c;
                B += lead;
                if (ag.first_tile >= 0) {
                    if (first_event < 0) { first_event = ag.first_event; base_first = B; }
                    B += ag.cnt_after_first;
                    const TileRec t = TR[k];
                    A.valid = 1; A.s = t.last_s; A.hi = t.hi; A.mid = t.mid; A.lo = t.lo;
                }
                E += ag.n_events;
            }
        }
        phase_sync();
    }
    if (lane == 0) {
        p25fe_result_t r;
        r.n_baseband = a.n_baseband;
        r.n_dibits = B;
        r.n_sync = E;
        r.anchor_out = A;
        r.first_event = first_event;
        r.n_dibits_after_first = first_event >= 0 ? B - base_first : 0;
        a.result[ch] = r;
    }
}

struct SliceArgs {
    const float* bb;
    long bb_stride;
    long n_hist;
    long n;
    long abs0;
    int n_tiles;
    const uint8_t* events;
    long ev_stride;
    const ScanOut* outs;
    const TileRec* recs;
    const GroupCarry* carries;          // [ch][n_groups]
    const unsigned long long* tsum;     // [ch][n_tiles] K2's packed summaries (0: no event in the tile)
    int n_groups;
    uint8_t* dibits;            // [ch][dibit_stride]
    long dibit_stride;
    int64_t* sync_pos;          // nullable
    uint64_t* sync_dibit;       // nullable
    long sync_stride;
};

// K4: slice the anchored symbol instants.  One wave per workgroup walks K4_SUBS consecutive tiles; a lane owns 16
// consecutive samples of a tile (read straight from global memory: the baseband and the flags were just written
// and are L2 / MALL resident), the next tile's samples are requested before the current ones are used.
//   * The carry-in of the workgroup's tiles (K3's ScanOut / GroupCarry, the anchor's TileRec) is fetched once, one
//     lane per tile, and broadcast per tile with v_readlane: two dependent loads per workgroup, not per tile.
//   * A tile WITHOUT events (7 of 8 in a P25 stream; K2's packed summary says so) needs no flags and no scan: the
//     instants and their ranks are closed forms of the carry-in anchor.  Without an anchor it is skipped unread.
#ifndef P25FE_K4_SUBS
#define P25FE_K4_SUBS 1
#endif
#ifndef P25FE_K4_WPS
#define P25FE_K4_WPS 4
#endif
constexpr int K4_SUBS = P25FE_K4_SUBS;
constexpr int K4_VPL = TB / WV;                                  // 16 samples per lane
static_assert(K4_VPL > SPS && K4_VPL <= 2 * SPS, "closed forms assume 1..2 instants per lane");

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ long rl_l(long v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffL), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned char slice_dibit(float v, float hi, float mid, float lo)
{
    return v >= hi ? 1 : v >= mid ? 0 : v >= lo ? 2 : 3;
}

__global__ __launch_bounds__(WV, P25FE_K4_WPS) void k_slice(SliceArgs a)
{
    const int lane = threadIdx.x, ch = blockIdx.y;
    const int tile_first = blockIdx.x * K4_SUBS;
    const float* bbp = a.bb + (size_t)ch * a.bb_stride;
    const bool aligned = ((reinterpret_cast<uintptr_t>(bbp) & 15u) == 0) && ((a.ev_stride & 15) == 0);

    // carry-in of tile tile_first + l, computed by lane l (lanes >= K4_SUBS repeat the last one)
    long c_anchor; float c_hi, c_mid, c_lo; int c_valid, c_has_ev; unsigned long long c_dibit_off; unsigned c_event_off, c_ph;
    {
        int tl = tile_first + (lane < K4_SUBS ? lane : K4_SUBS - 1);
        tl = tl < a.n_tiles ? tl : a.n_tiles - 1;
        const long t0 = (long)tl * TB;
        const ScanOut so = a.outs[(size_t)ch * a.n_tiles + tl];
        const GroupCarry gc = a.carries[(size_t)ch * a.n_groups + tl / K3_GROUP];
        c_has_ev = a.tsum[(size_t)ch * a.n_tiles + tl] != 0ull;
        c_event_off = (unsigned)(gc.event_base + so.event_off);
        if (so.src >= 0) {                                   // an event of this group governs the tile's start
            const TileRec t = a.recs[(size_t)ch * a.n_tiles + so.src];
            c_anchor = t.last_s; c_hi = t.hi; c_mid = t.mid; c_lo = t.lo; c_valid = 1;
            c_dibit_off = gc.base_after_first + so.dibit_off;
        } else {                                             // the group's carry-in governs it: closed-form count so far
            c_anchor = gc.anchor_s; c_hi = gc.hi; c_mid = gc.mid; c_lo = gc.lo; c_valid = gc.valid;
            const long glo = a.abs0 + (long)(tl / K3_GROUP) * K3_GROUP * TB;
            c_dibit_off = gc.dibit_base + (gc.valid ? (unsigned long long)count_instants(gc.anchor_s, glo, a.abs0 + t0) : 0ull);
        }
        // distance of the tile's first sample to the carry-in anchor, mod 10 (the one 64-bit modulo)
        c_ph = c_valid ? (unsigned)((a.abs0 + t0 - c_anchor) % SPS) : 0u;
    }

    float4 nb[4];
    uint4 nev = make_uint4(0u, 0u, 0u, 0u);
    // request tile t's samples (and flags if it has events); whole, aligned tiles only -- others are read element-wise
    auto request = [&](int t, int has_ev, int valid) {
        const long t0 = (long)t * TB;
        const bool fast = aligned && t < a.n_tiles && a.n - t0 >= TB;
        if (fast && (has_ev || valid)) {
            // with events: a lane's 16 consecutive samples; without: lane-interleaved vectors (contiguous 1 KB per load)
            const float4* bp = reinterpret_cast<const float4*>(bbp + t0) + (has_ev ? lane * 4 : lane);
            const int qs = has_ev ? 1 : WV;
#pragma unroll
            for (int q = 0; q < 4; ++q) nb[q] = bp[q * qs];
            if (has_ev) nev = *reinterpret_cast<const uint4*>(a.events + (size_t)ch * a.ev_stride + t0 + lane * K4_VPL);
        }
    };
    request(tile_first, rl_i(c_has_ev, 0), rl_i(c_valid, 0));

#pragma unroll 1
    for (int it = 0; it < K4_SUBS; ++it) {
        const int tile = tile_first + it;
        if (tile >= a.n_tiles) break;                               // uniform
        const long t0 = (long)tile * TB;
        const int tn = (a.n - t0 < TB) ? (int)(a.n - t0) : TB;
        const int has_ev = rl_i(c_has_ev, it), valid = rl_i(c_valid, it);
        const bool fast = aligned && tn == TB;
        if (fast && !has_ev) {
            // no event in the tile: instants at i = f0 + 10 m under the carry-in anchor, ranks in closed form.  The lane
            // holds four groups of 4 consecutive samples (interleaved vectors), each with at most one instant (4 < 10).
            float4 cb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cb[q] = nb[q];
            if (it + 1 < K4_SUBS) request(tile + 1, rl_i(c_has_ev, it + 1), rl_i(c_valid, it + 1));
            if (!valid) continue;                                   // nothing decided yet: no instants
            const float hi0 = rl_f(c_hi, it), mid0 = rl_f(c_mid, it), lo0 = rl_f(c_lo, it);
            const unsigned base_ph = (unsigned)rl_i((int)c_ph, it);
            uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride + (unsigned long long)rl_l((long)c_dibit_off, it);
            const int f0 = (int)((SPS - base_ph) % (unsigned)SPS);  // first instant of the tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = 4 * (lane + q * WV);
                const unsigned ph = (base_ph + (unsigned)g) % (unsigned)SPS;
                const int f = (int)((SPS - ph) % (unsigned)SPS);
#if defined(P25FE_ABLATE4) && P25FE_ABLATE4 == 2     // measurement build: fast-path stores suppressed
                if (f < 4 && cb[q].x == 123.f) {
#else
                if (f < 4) {
#endif
                    const float v = f == 0 ? cb[q].x : f == 1 ? cb[q].y : f == 2 ? cb[q].z : cb[q].w;
                    out[(unsigned)(g + f - f0) / (unsigned)SPS] = slice_dibit(v, hi0, mid0, lo0);
                }
            }
            continue;
        }
        float bv[K4_VPL];
        unsigned evw[4] = {0u, 0u, 0u, 0u};
        if (fast) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { bv[4 * q] = nb[q].x; bv[4 * q + 1] = nb[q].y; bv[4 * q + 2] = nb[q].z; bv[4 * q + 3] = nb[q].w; }
            evw[0] = nev.x; evw[1] = nev.y; evw[2] = nev.z; evw[3] = nev.w;
        } else {
            const float* bp = bbp + t0 + lane * K4_VPL;
            const uint8_t* evp = a.events + (size_t)ch * a.ev_stride + t0 + lane * K4_VPL;
#pragma unroll
            for (int u = 0; u < K4_VPL; ++u) {
                const bool in = lane * K4_VPL + u < tn;
                bv[u] = in ? bp[u] : 0.f;
                if (has_ev && in && evp[u]) evw[u >> 2] |= 1u << (8 * (u & 3));
            }
        }
        if (it + 1 < K4_SUBS) request(tile + 1, rl_i(c_has_ev, it + 1), rl_i(c_valid, it + 1));
        if (!has_ev && !valid) continue;                            // nothing decided yet: no instants

        const float hi0 = rl_f(c_hi, it), mid0 = rl_f(c_mid, it), lo0 = rl_f(c_lo, it);
        const unsigned base_ph = (unsigned)rl_i((int)c_ph, it);
        const unsigned long long dibit_off = (unsigned long long)rl_l((long)c_dibit_off, it);
        uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride + dibit_off;
        const int i0 = lane * K4_VPL;

#if defined(P25FE_ABLATE4) && P25FE_ABLATE4 == 1     // measurement build: tiles with events skipped
        if (bv[0] != 123.f) continue;
#endif
        // general tile: events inside.  Latest own event before each lane's first sample, count, rank, emit.
        int my_last = -1, my_ev = 0;
#pragma unroll
        for (int u = 0; u < K4_VPL; ++u)
            if ((evw[u >> 2] >> (8 * (u & 3))) & 1u) { my_last = i0 + u; ++my_ev; }
        const int incl = wave_incl_max_i(my_last, lane);
        int incoming = __shfl_up(incl, 1, 64);
        if (lane == 0) incoming = -1;
        int cur = incoming;
        int cnt = 0;
        unsigned inst = 0;
        const unsigned cph = (base_ph + (unsigned)i0) % (unsigned)SPS;
#pragma unroll
        for (int u = 0; u < K4_VPL; ++u) {
            const int i = i0 + u;
            if (i < tn) {
                bool is = false;
                if (cur >= 0) is = ((unsigned)(i - (cur - W)) % (unsigned)SPS) == 0u;
                else if (valid) is = ((cph + (unsigned)u) % (unsigned)SPS) == 0u;
                if (is) { ++cnt; inst |= 1u << u; }
                if ((evw[u >> 2] >> (8 * (u & 3))) & 1u) cur = i;
            }
        }
        // one wave scan for both ranks: instants in the low half, events in the high half (each <= 1024 per tile)
        const int mine = cnt | (my_ev << 16);
        const int packed = wave_incl_sum(mine, lane) - mine;
        int rank = packed & 0xffff;
        int evrank = packed >> 16;
        const unsigned event_off = (unsigned)rl_i((int)c_event_off, it);
        cur = incoming;
        int thr_for = -2;                      // tile-local anchor whose thresholds are cached in hi / mid / lo
        float hi = hi0, mid = mid0, lo = lo0;
        // K2 left the thresholds of the tile's LAST event in its TileRec: with one event per tile (the normal case)
        // no lane recomputes anything; only earlier events of a multi-event tile take the 24-load path below.
        const TileRec own = a.recs[(size_t)ch * a.n_tiles + tile];
        const int own_last = (int)(own.last_s - a.abs0 - t0) + W;
#pragma unroll
        for (int u = 0; u < K4_VPL; ++u) {
            const int i = i0 + u;
            if (i < tn) {
                // the instant at index i (if any) is governed by events decided before i
                if ((inst >> u) & 1u) {
                    if (cur >= 0 && cur != thr_for && cur == own_last) {
                        hi = own.hi; mid = own.mid; lo = own.lo;
                        thr_for = cur;
                    }
                    if (cur >= 0 && cur != thr_for) {
                        // thresholds of an in-tile anchor: same arithmetic as K2 (same bits), from global memory
                        sync_thresholds_global(bbp, a.n_hist, a.n, t0 + cur - W, hi, mid, lo);
                        thr_for = cur;
                    }
                    out[rank++] = slice_dibit(bv[u], hi, mid, lo);
                }
                if ((evw[u >> 2] >> (8 * (u & 3))) & 1u) {
                    if (a.sync_pos && (long)(event_off + evrank) < a.sync_stride) {
                        a.sync_pos[(size_t)ch * a.sync_stride + event_off + evrank] = a.abs0 + t0 + i - W;
                        // index of the first dibit this detection governs = dibits for instants <= i
                        a.sync_dibit[(size_t)ch * a.sync_stride + event_off + evrank] = dibit_off + rank;
                    }
                    ++evrank;
                    cur = i;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K5: network identifier after each frame sync (MessageEvent::PacketNID, src/recv.rs:216-222).
// One workgroup per sync event gathers the 32 NID dibits (status symbol at +11 skipped) and finds the nearest of the
// 65536 BCH(63,16,23) code words by exhaustive Hamming search (256 per thread, two 256-entry XOR tables in LDS) --
// equivalent to a bounded-distance decoder when at most t = 11 bits are wrong, and embarrassingly parallel.
// ------------------------------------------------------------------------------------------
// Batch form: blockIdx.y = channel, per-channel strides, event / dibit counts read from the channel's p25fe_result_t.
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

// Per-channel observability record (src/hub.rs:344, 401-402, 557-581): one workgroup per
// channel folds the channel's NID records into the "bch" CodeStats row and copies the lock state.
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
        o.locked = r.anchor_out.valid;
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

// Carry resolution across time shards (BASELINE.json config 5): the same "latest anchor wins" rule as K3, one level
// up.  summaries[r] were produced assuming no carry-in; shard r's carry-in is the anchor_out of the latest earlier
// shard that has an event of its own, and its dibit offset adds the closed-form count of instants that the carry-in
// governs before the shard's first own event.  Shared by the host entry point and the one-thread device kernel.
__host__ __device__ inline void shard_resolve_impl(const p25fe_result_t* summaries, const uint64_t* shard_bb0,
                                                    const uint64_t* shard_bb_n, int n_shards, p25fe_anchor_t* anchor_in,
                                                    uint64_t* dibit_offset)
{
    p25fe_anchor_t cur;
    cur.s = 0; cur.hi = cur.mid = cur.lo = 0.f; cur.valid = 0;
    uint64_t off = 0;
    for (int r = 0; r < n_shards; ++r) {
        anchor_in[r] = cur;
        dibit_offset[r] = off;
        const long lo = (long)shard_bb0[r], hi = (long)(shard_bb0[r] + shard_bb_n[r]);
        const long pre_hi = summaries[r].first_event >= 0 ? (long)summaries[r].first_event + 1 : hi;
        const uint64_t pre = cur.valid ? (uint64_t)count_instants(cur.s, lo, pre_hi) : 0;
        off += pre + (summaries[r].first_event >= 0 ? summaries[r].n_dibits_after_first : 0);
        if (summaries[r].first_event >= 0) cur = summaries[r].anchor_out;
    }
}

__global__ void k_shard_resolve(const p25fe_result_t* summaries, const uint64_t* shard_bb0, const uint64_t* shard_bb_n,
                                int n_shards, p25fe_anchor_t* anchor_in, uint64_t* dibit_offset)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        shard_resolve_impl(summaries, shard_bb0, shard_bb_n, n_shards, anchor_in, dibit_offset);
}

}  // namespace p25k
