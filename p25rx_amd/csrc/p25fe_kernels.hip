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
//
// Arithmetic contract (docs/SPEC.md section 3): fp32, fma only where the spec says fma, single
// accumulator per output in tap order 0..T-1.  Compiled with -ffp-contract=off.
// Stencil / element-wise work: no MFMA.  HBM-bound by design: 8 B in + 0.8 B out per IQ sample.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p25fe.h"
#include "p25fe_spec.h"

namespace p25k {

// ------------------------------------------------------------------------------------------
// geometry of K1, parametrised by the workgroup size NTK (64 = one wave per workgroup: no
// s_barrier at all, eight independent waves per CU interleave their phases freely)
// ------------------------------------------------------------------------------------------
constexpr int DEC = P25FE_DECIM;
constexpr int T1 = P25FE_T1;
constexpr int T2 = P25FE_T2;
constexpr int BOX = P25FE_BOXCAR;
constexpr int HALO_Y = BOX;                  // y needed from m0-10 (fm needs y[m-1], boxcar needs fm[m-9])
constexpr int HALO_D = HALO_Y + (T2 - 1);    // 50: d needed from m0-50
constexpr int F_CARRY = 12;                  // fm carry (9 used)
constexpr int Y_CARRY = 2;                   // y carry (1 used), keeps 16-B alignment
constexpr int D_CARRY = T2 - 1;              // d carry
// history (input samples before the first owned one) needed for exact results
constexpr int HIST_IQ = DEC * HALO_D + (T1 - 1) + (DEC - 1);   // 284

// PK = consecutive FIR outputs per thread (odd: lane stride 2*5*PK / 2*PK dwords -> conflict-free ds_read_b64)
template <int NTK, int PK> struct Geo {
    static constexpr int NT = NTK;
    static constexpr int P = PK;
    static constexpr int SUB = NTK * PK;                  // decimated samples per sub-tile
    static constexpr int XWIN = DEC * SUB + (T1 - DEC);   // input samples feeding one sub-tile of d
    static constexpr int XIN_N = 8 + (XWIN + 8 + 7) / 8 * 8;   // 8 spare entries in front + window + loader slack
    static constexpr int D_N = D_CARRY + SUB;
    static constexpr int Y_N = Y_CARRY + SUB;
    static constexpr int F_N = F_CARRY + SUB;
    static constexpr size_t LDS_BYTES = sizeof(float2) * (XIN_N + D_N + Y_N) + sizeof(float) * (F_N + T1 + T2 + 3);
    static constexpr int WAVES_PER_SIMD = PK <= 3 ? 3 : 2;   // register budget the kernel is compiled for
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

// SPEC 3.3: FM discriminator, angle of s * conj(prev) times fs / (2 pi dev)   (src/demod.rs:54, 110)
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
#ifdef P25FE_LDS_READ2
    return *p;
#else
    typedef const volatile unsigned long long __attribute__((address_space(3))) * lds_u64_ptr;
    const unsigned long long u = *((lds_u64_ptr)p);                 // generic -> LDS address space: stays a DS op
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
#endif
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
template <int FMT, int NTK, int PK> struct Loader {
    using G = Geo<NTK, PK>;
    static constexpr int LOG_SPV = FMT == P25FE_FMT_CF32 ? 1 : 3;   // samples per 16-B vector: 2 (cf32) or 8 (u8 pairs)
    static constexpr int SPV = 1 << LOG_SPV;
    static constexpr int NV = (G::XWIN + SPV + SPV * NTK - 1) / (SPV * NTK);   // vectors per thread: 13 / 4
    uint4 v[NV];

    // base: pointer to owned sample 0 of this channel; first: index of the first window sample
    // i_last: last sample index this workgroup will ever need (loads past it collapse onto one cached vector)
    __device__ __forceinline__ void load(const void* base, long first, long n_hist, long n_new, long i_last, int tid)
    {
        const uint4* p = reinterpret_cast<const uint4*>(base);
        const long vfirst = (first >> LOG_SPV) + tid;               // floor division (arithmetic shift)
        const long vlo = (-n_hist) >> LOG_SPV;                      // vector holding sample -n_hist
        const long last = i_last < n_new - 1 ? i_last : n_new - 1;
        const long vhi = last >> LOG_SPV;                           // vector holding the last useful sample
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            long vi = vfirst + (long)j * NTK;
            vi = vi < vlo ? vlo : vi;
            vi = vi > vhi ? vhi : vi;
            v[j] = p[vi];
        }
    }

    // XIN[k] must hold sample first + k; vector j of thread tid holds samples SPV * (vfirst + j * NTK) + e
    __device__ __forceinline__ void store(float2* XIN, long first, long n_hist, long n_new, int tid) const
    {
        const long first_al = (first >> LOG_SPV) << LOG_SPV;
        const int sh = (int)(first - first_al);                     // 0 .. SPV-1
        float2* XS = XIN - sh;                                      // XIN is preceded by SPV spare entries
        const bool interior = first_al >= -n_hist && first_al + (long)NV * SPV * NTK <= n_new;   // uniform
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int e = 0; e < SPV; ++e) {
                const int k = SPV * (tid + j * NTK) + e;             // always inside the XIN allocation
                float2 s;
                if (FMT == P25FE_FMT_CF32) {
                    s = make_float2(__uint_as_float(w[2 * (e & 1)]), __uint_as_float(w[2 * (e & 1) + 1]));
                } else {
                    const unsigned pair = (w[(e >> 1) & 3] >> (16 * (e & 1))) & 0xffffu;
                    s = make_float2(u8_to_f32(pair & 0xffu), u8_to_f32(pair >> 8));   // low byte = I (SPEC 3.1)
                }
                if (!interior) {
                    const long i = first_al + SPV * (long)(tid + j * NTK) + e;
                    if (i < -n_hist || i >= n_new) s = make_float2(0.f, 0.f);
                }
                // only the last round of vectors can run past the window: everything else is stored unconditionally
                if (j < NV - 1 || k < G::XIN_N - 8) XS[k] = s;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// K1: fused front end.  One workgroup walks `subs_per_seg` consecutive sub-tiles of one channel;
// FIR / FM / boxcar context is carried in LDS between sub-tiles, so the (T2-1)+10 sample halo is
// recomputed once per segment only.  The next sub-tile's input window is prefetched into
// registers while the current one is processed.
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

// Phase boundary inside a workgroup.  With NTK == 64 the workgroup is one wave: LDS operations of
// one wave execute in order, so only the compiler must be kept from reordering across the
// boundary and the hardware barrier disappears.
template <int NTK> __device__ __forceinline__ void phase_sync()
{
    if (NTK > 64) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// CT = true: the handle's taps are the build's default tables (p25fe_spec.h) -> immediates, no
// registers or LDS reads spent on coefficients.  CT = false: caller-supplied taps, broadcast-read from LDS.
template <int FMT, bool CT, int NTK, int PK>
__global__ __launch_bounds__(NTK, (Geo<NTK, PK>::WAVES_PER_SIMD)) void k_frontend(K1Args a, const Taps* __restrict__ gtaps)
{
    using G = Geo<NTK, PK>;
    constexpr int SUB = G::SUB;
    constexpr int P = PK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* XIN = reinterpret_cast<float2*>(smem) + 8;              // 8 spare entries in front (loader shift)
    float2* D = reinterpret_cast<float2*>(smem) + G::XIN_N;
    float2* Y = D + G::D_N;
    float* F = reinterpret_cast<float*>(Y + G::Y_N);
    float* TAPS = F + G::F_N;                                       // [T1 | T2], only when !CT
    const int tid = threadIdx.x;
    if (!CT) {
        for (int k = tid; k < T1; k += NTK) TAPS[k] = gtaps->dec[k];
        for (int k = tid; k < T2; k += NTK) TAPS[T1 + k] = gtaps->ch[k];
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

    // zero the carries (their garbage would only reach never-stored outputs, but keep it tidy)
    for (int k = tid; k < D_CARRY; k += NTK) D[k] = make_float2(0.f, 0.f);
    if (tid < Y_CARRY) Y[tid] = make_float2(0.f, 0.f);
    if (tid < F_CARRY) F[tid] = 0.f;

    Loader<FMT, NTK, PK> ld;
    long dlo = m_seg0 - HALO_D;                                    // first d index of this sub-tile
    const long i_last = (long)a.o0 + DEC * (m_seg1 - 1);          // newest input sample this segment needs
    ld.load(xb, (long)a.o0 + DEC * dlo - (T1 - 1), a.n_hist, a.n_new, i_last, tid);
    float pw = 0.f;
    // Outputs are kept in registers for one sub-tile and stored at the top of the next one, BEFORE the
    // prefetch loads are issued: vmcnt counts loads and stores in one in-order queue, so stores issued
    // after the prefetch would force the wait at the top of the loop to drain them too (measured: the
    // loop then ran at the HBM write round-trip per sub-tile).
    float outv[P];
#pragma unroll
    for (int q = 0; q < P; ++q) outv[q] = 0.f;
    long out_lo = m_seg0 - 2 * (long)SUB;                           // nothing in range yet
    auto flush_outputs = [&]() {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const long m = out_lo + tid + q * NTK;
            if (m >= m_seg0 && m < m_seg1) bb[m] = outv[q];
        }
    };

    for (int it = 0; it < a.subs_per_seg; ++it, dlo += SUB) {
        if (dlo >= m_seg1) break;                                  // uniform
        const long first = (long)a.o0 + DEC * dlo - (T1 - 1);      // XIN[k] = x[first + k]
        ld.store(XIN, first, a.n_hist, a.n_new, tid);
        phase_sync<NTK>();
        flush_outputs();                                            // previous sub-tile's outputs (lane-predicated)
        // unconditional prefetch: past the segment's end the clamp makes every lane read one cached vector
        ld.load(xb, first + (long)DEC * SUB, a.n_hist, a.n_new, i_last, tid);

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 1      // measurement builds only (tools/ablate.sh): stop after the load pipeline
        continue;
#endif
        // ---- stage 2: 5:1 decimating FIR (src/demod.rs:87). Thread: d[dlo + 5 tid + p], p = 0..4.
        // Output p needs x[first + 5(5 tid + p) + (T1-1) - k], k = 0..T1-1  -> XIN[25 tid + 5p + 30 - k].
        {
            const float2* w = XIN + (DEC * P) * tid;
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
#pragma unroll
            for (int p = 0; p < P; ++p) D[D_CARRY + P * tid + p] = acc[p];
        }
        phase_sync<NTK>();

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 2
        continue;
#endif
        // ---- stage 3: channel FIR (src/demod.rs:93). Thread: y[dlo + 5 tid + p] from D[5 tid + p + 40 - k].
        {
            const float2* w = D + P * tid;
            float2 acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = (P - 1) + T2 - 1; j >= 0; --j) {
                const float2 s = lds_read_c(w + j);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = p + (T2 - 1) - j;
                    if (k >= 0 && k < T2) {
                        acc[p].x = __builtin_fmaf(tap_ch(k), s.x, acc[p].x);
                        acc[p].y = __builtin_fmaf(tap_ch(k), s.y, acc[p].y);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                Y[Y_CARRY + P * tid + p] = acc[p];
                const long m = dlo + P * tid + p;
                if (a.power_partial && m >= m_seg0 && m < m_seg1 && m >= 0) {
                    const float q0 = acc[p].x * acc[p].x, q1 = acc[p].y * acc[p].y;
                    pw = pw + (q0 + q1);
                }
            }
        }
        phase_sync<NTK>();

#if defined(P25FE_ABLATE) && P25FE_ABLATE <= 3
        continue;
#endif
        // ---- stage 4: FM discriminator (src/demod.rs:109-111), one output per thread per pass
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int i = tid + q * NTK;
            F[F_CARRY + i] = fm_discriminate(Y[Y_CARRY + i], Y[Y_CARRY + i - 1]);
        }
        phase_sync<NTK>();

    #if defined(P25FE_ABLATE) && P25FE_ABLATE <= 4
        continue;
#endif
        // ---- stage 5: boxcar (src/demod.rs:114); lane-consecutive outputs -> coalesced stores (deferred)
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int i = tid + q * NTK;
            const float* f = F + F_CARRY + i;
            float acc = f[0];
#pragma unroll
            for (int j = 1; j < BOX; ++j) acc = acc + f[-j];
            outv[q] = acc * P25FE_BOXCAR_SCALE;
        }
        out_lo = dlo;
        phase_sync<NTK>();

        // ---- carry context to the next sub-tile
        for (int k = tid; k < D_CARRY; k += NTK) D[k] = D[SUB + k];
        if (tid < Y_CARRY) Y[tid] = Y[SUB + tid];
        if (tid < F_CARRY) F[tid] = F[SUB + tid];
        // ordered before their next use by the phase boundary that follows the next XIN store
    }
    flush_outputs();

    if (a.power_partial) {
        // block reduction of the |y|^2 partials (tree; tolerance vs the sequential fold is in the tests)
        phase_sync<NTK>();
        float* red = F;
        red[tid] = pw;
        phase_sync<NTK>();
        for (int s = NTK / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = red[tid] + red[tid + s];
            phase_sync<NTK>();
        }
        if (tid == 0) a.power_partial[(size_t)ch * gridDim.x + blockIdx.x] = red[0];
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
// K2..K4: symbol receiver as a scan (SPEC 3.6-3.8)
//
// A detection at s (sync word's last symbol) is DECIDED at e = s + W (the peak window is complete)
// and governs instants n > e: instant n is
// governed by the detection with the latest e <= n.  Tiles own the EVENTS whose e falls in
// them, so every dependency points left.
// ------------------------------------------------------------------------------------------
constexpr int W = P25FE_PEAK_W;
constexpr int SPS = P25FE_SPS;
constexpr int SYNC_SPAN = P25FE_SYNC_SPAN;                   // 230
constexpr int TB = 2048;                                     // baseband samples per tile
constexpr int NT = 256;                                      // threads per workgroup in K2 / K4
constexpr int VPT = TB / NT;                                 // 8 consecutive samples per thread
constexpr int HIST_BB = SYNC_SPAN + 2 * W;                   // 240: left context of a tile
constexpr int BT_N = TB + HIST_BB + 3;                       // LDS baseband tile (+pad)
constexpr int CT_N = TB + 2 * W + 2;                         // c[] for s in [a-2W-1, a+TB-W-1) plus peak lookahead

struct TileRec {            // per (channel, tile) summary written by K2
    long first_event;       // absolute decision index e = s + W of the tile's first event, -1 if none
    long last_s;            // s of the tile's last event (absolute), valid if first_event >= 0
    float hi, mid, lo;      // thresholds of the last event
    int n_events;
    long post_count;        // instants in (first_event, tile_end) under the tile's own events
};

struct ScanOut {            // per (channel, tile) carry-in written by K3
    long anchor_s;
    float hi, mid, lo;
    int valid;
    unsigned long long dibit_off;   // dibits of the range before this tile
    unsigned long long event_off;   // events of the range before this tile
};

__device__ __forceinline__ float bb_at(const float* bbp, long n_hist, long n, long i)
{
    return (i >= -n_hist && i < n) ? bbp[i] : 0.0f;
}

// thresholds from the sync word ending at local LDS index `is` (SPEC 3.7)
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

// Block-wide exclusive scans over per-thread (last event position, count).  scratch: >= 8 longs + 8 ints.
__device__ __forceinline__ long block_excl_max(long v, long* sh, int tid)
{
    const int lane = tid & 63, wv = tid >> 6;
    const long inc = wave_incl_max(v, lane);
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    long prev = __shfl_up(inc, 1, 64);
    if (lane == 0) prev = -1;
    long carry = -1;
    for (int k = 0; k < wv; ++k) carry = sh[k] > carry ? sh[k] : carry;
    __syncthreads();
    return prev > carry ? prev : carry;
}
__device__ __forceinline__ int block_excl_sum(int v, int* sh, int tid, int& total)
{
    const int lane = tid & 63, wv = tid >> 6;
    const int inc = wave_incl_sum(v, lane);
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int carry = 0;
    for (int k = 0; k < wv; ++k) carry += sh[k];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return carry + inc - v;
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
};

// K2: correlate, peak-pick, flag events, summarise the tile.
__global__ __launch_bounds__(NT) void k_sync(SyncArgs a)
{
    __shared__ float BT[BT_N];          // BT[k] = b[t0 - HIST_BB + k]
    __shared__ float CT[CT_N];          // CT[k] = c[t0 - 2W + k]
    __shared__ uint8_t CAND[CT_N];
    __shared__ long shl[8];
    __shared__ int shi[8];

    const int tid = threadIdx.x, tile = blockIdx.x, ch = blockIdx.y;
    const long t0 = (long)tile * TB;                            // local index of the tile's first sample
    const float* bbp = a.bb + (size_t)ch * a.bb_stride;
    const long tn = (a.n - t0 < TB) ? a.n - t0 : TB;            // samples in this tile

    for (int k = tid; k < BT_N; k += NT) BT[k] = bb_at(bbp, a.n_hist, a.n, t0 - HIST_BB + k);
    __syncthreads();

    // c[s], cand[s] for s = t0 - 2W + k, k in [0, TB + 2W): needs b[s - 230 .. s]
    for (int k = tid; k < TB + 2 * W; k += NT) {
        // s = t0 - 2W + k  ->  BT index of b[s] = s - (t0 - HIST_BB) = k + HIST_BB - 2W = k + SYNC_SPAN
        const int is = k + SYNC_SPAN;
        float c = 0.f, e = 0.f;
#pragma unroll
        for (int j = 0; j < P25FE_SYNC_DIBITS; ++j) {
            const float v = BT[is - SPS * (P25FE_SYNC_DIBITS - 1 - j)];
            c = ((P25FE_SYNC_SIGN_MASK >> j) & 1u) ? c + v : c - v;
            e = __builtin_fmaf(v, v, e);
        }
        CT[k] = c;
        CAND[k] = (c > 0.0f) && (e >= P25FE_SYNC_E_MIN) && (c * c >= P25FE_SYNC_RHO2_N * e);
    }
    __syncthreads();

    // event at local index i (decided when sample t0 + i arrives) <=> detection at s = t0 + i - W -> CT index k = i + W.
    // s must lie inside the stream so far: s >= -n_hist (older samples read as zero anyway) and the peak
    // window s + W = t0 + i - 1 < n is guaranteed by i < tn.
    long my_last = -1;
    int my_ev = 0;
    uint8_t evl[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid * VPT + u;
        const int k = i + W;
        bool det = false;
        if (i < tn && CAND[k]) {
            const float cm = CT[k];
            det = true;
#pragma unroll
            for (int d = 1; d <= W; ++d) det = det && (cm > CT[k - d]) && (cm >= CT[k + d]);
        }
        evl[u] = det;
        if (det) { my_last = t0 + i; ++my_ev; }
    }
    if (a.events) {
        uint8_t* evp = a.events + (size_t)ch * a.ev_stride + t0;
        // 8 consecutive bytes per thread: one 8-byte store when aligned (t0 multiple of 2048, stride multiple of 8)
        if (tid * VPT + VPT <= tn && ((a.ev_stride & 7) == 0)) {
            uint2 pk;
            pk.x = evl[0] | (evl[1] << 8) | (evl[2] << 16) | ((unsigned)evl[3] << 24);
            pk.y = evl[4] | (evl[5] << 8) | (evl[6] << 16) | ((unsigned)evl[7] << 24);
            *reinterpret_cast<uint2*>(evp + tid * VPT) = pk;
        } else {
#pragma unroll
            for (int u = 0; u < VPT; ++u)
                if (tid * VPT + u < tn) evp[tid * VPT + u] = evl[u];
        }
    }

    // latest own event before each thread's first sample, then count instants under own events
    const long incoming = block_excl_max(my_last, shl, tid);
    long cur = incoming;        // local e-position of the governing event, -1: carry-in (unknown here)
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid * VPT + u;
        if (i < tn) {
            if (cur >= 0) {                             // events decided BEFORE this sample govern it
                const long s = cur - W;                 // local index of the anchor
                if ((t0 + i - s) % SPS == 0) ++cnt;     // t0 + i > s always
            }
            if (evl[u]) cur = t0 + i;
        }
    }
    int total_cnt, total_ev;
    block_excl_sum(cnt, shi, tid, total_cnt);
    block_excl_sum(my_ev, shi, tid, total_ev);
    // first / last event of the tile
    __shared__ long first_ev, last_ev;
    if (tid == 0) first_ev = -1;
    if (tid == NT - 1) last_ev = incoming > my_last ? incoming : my_last;
    __syncthreads();
    if (my_last >= 0 && incoming < 0) {                        // the one thread whose event has none before it
        long f = -1;
#pragma unroll
        for (int u = VPT - 1; u >= 0; --u) if (evl[u]) f = t0 + tid * VPT + u;
        first_ev = f;
    }
    __syncthreads();
    if (tid == 0) {
        TileRec r;
        r.first_event = first_ev >= 0 ? first_ev + a.abs0 : -1;
        r.n_events = total_ev;
        r.post_count = total_cnt;
        r.last_s = -1; r.hi = r.mid = r.lo = 0.f;
        if (last_ev >= 0) {
            const long s = last_ev - W;                         // local
            float hi, mid, lo;
            sync_thresholds(BT, (int)(s - (t0 - HIST_BB)), hi, mid, lo);
            r.last_s = s + a.abs0; r.hi = hi; r.mid = mid; r.lo = lo;
        }
        a.recs[(size_t)ch * a.n_tiles + tile] = r;
    }
}

struct ScanArgs {
    const TileRec* recs;
    ScanOut* outs;
    int n_tiles;
    long n;                 // owned samples per channel
    long abs0;
    const p25fe_anchor_t* anchor_in;    // nullable, [ch]
    p25fe_result_t* result;             // [ch]
    unsigned long long n_baseband;      // to report
};

// K3: one workgroup per channel.  Exclusive "latest anchor" scan over tiles, then dibit/event offsets.
__global__ __launch_bounds__(1024) void k_scan(ScanArgs a)
{
    __shared__ long sh_src[1024];
    __shared__ unsigned long long sh_cnt[1024];
    __shared__ unsigned long long sh_ev[1024];
    const int tid = threadIdx.x, ch = blockIdx.x;
    const TileRec* recs = a.recs + (size_t)ch * a.n_tiles;
    ScanOut* outs = a.outs + (size_t)ch * a.n_tiles;
    const int per = (a.n_tiles + 1023) / 1024;
    const int i0 = tid * per, i1 = (i0 + per < a.n_tiles) ? i0 + per : a.n_tiles;

    // pass A: index of the latest tile with an event, exclusive over tiles
    long last = -1;
    for (int i = i0; i < i1; ++i) if (recs[i].first_event >= 0) last = i;
    sh_src[tid] = last;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {               // Hillis-Steele inclusive max
        long v = sh_src[tid];
        if (tid >= d) { const long o = sh_src[tid - d]; v = o > v ? o : v; }
        __syncthreads();
        sh_src[tid] = v;
        __syncthreads();
    }
    long src = tid > 0 ? sh_src[tid - 1] : -1;         // latest event tile before my chunk
    p25fe_anchor_t ain;
    ain.valid = 0; ain.s = 0; ain.hi = ain.mid = ain.lo = 0.f;
    if (a.anchor_in) ain = a.anchor_in[ch];

    // pass B: per-tile carry-in anchor and counts
    unsigned long long my_cnt = 0, my_ev = 0;
    for (int i = i0; i < i1; ++i) {
        ScanOut o;
        if (src >= 0) { o.valid = 1; o.anchor_s = recs[src].last_s; o.hi = recs[src].hi; o.mid = recs[src].mid; o.lo = recs[src].lo; }
        else { o.valid = ain.valid; o.anchor_s = ain.s; o.hi = ain.hi; o.mid = ain.mid; o.lo = ain.lo; }
        const long tlo = a.abs0 + (long)i * TB;
        long thi = tlo + TB;
        if (thi > a.abs0 + a.n) thi = a.abs0 + a.n;
        const long pre_hi = recs[i].first_event >= 0 ? recs[i].first_event + 1 : thi;   // instant AT e is still the old anchor's
        const unsigned long long pre = o.valid ? (unsigned long long)count_instants(o.anchor_s, tlo, pre_hi) : 0ull;
        o.dibit_off = my_cnt;          // relative to my chunk for now
        o.event_off = my_ev;
        outs[i] = o;
        my_cnt += pre + (unsigned long long)recs[i].post_count;
        my_ev += (unsigned long long)recs[i].n_events;
        if (recs[i].first_event >= 0) src = i;
    }
    sh_cnt[tid] = my_cnt;
    sh_ev[tid] = my_ev;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned long long v = sh_cnt[tid], w = sh_ev[tid];
        if (tid >= d) { v += sh_cnt[tid - d]; w += sh_ev[tid - d]; }
        __syncthreads();
        sh_cnt[tid] = v; sh_ev[tid] = w;
        __syncthreads();
    }
    const unsigned long long base_cnt = tid > 0 ? sh_cnt[tid - 1] : 0ull;
    const unsigned long long base_ev = tid > 0 ? sh_ev[tid - 1] : 0ull;
    for (int i = i0; i < i1; ++i) { outs[i].dibit_off += base_cnt; outs[i].event_off += base_ev; }

    if (tid == 1023) {
        p25fe_result_t r;
        r.n_baseband = a.n_baseband;
        r.n_dibits = sh_cnt[1023];
        r.n_sync = sh_ev[1023];
        const long lt = sh_src[1023];
        if (lt >= 0) { r.anchor_out.valid = 1; r.anchor_out.s = recs[lt].last_s; r.anchor_out.hi = recs[lt].hi; r.anchor_out.mid = recs[lt].mid; r.anchor_out.lo = recs[lt].lo; }
        else r.anchor_out = ain;
        // summary for time-sharding: first own event and the dibits governed by own events
        long fe = -1; unsigned long long after = 0;
        r.first_event = fe; r.n_dibits_after_first = after;
        a.result[ch] = r;
    }
    // first_event / n_dibits_after_first: first tile with an event (min-reduce), then suffix counts
    __syncthreads();
    long ft = 0x7fffffffffffffffL;
    for (int i = i0; i < i1; ++i) if (recs[i].first_event >= 0) { ft = i; break; }
    sh_src[tid] = ft;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (tid < d) { const long o = sh_src[tid + d]; if (o < sh_src[tid]) sh_src[tid] = o; }
        __syncthreads();
    }
    if (tid == 0 && sh_src[0] != 0x7fffffffffffffffL) {
        const long f = sh_src[0];
        // dibits after the first event = total - (dibits before tile f) - (pre-count of tile f)
        const ScanOut of = outs[f];
        const long tlo = a.abs0 + f * TB;
        const unsigned long long pre = of.valid ? (unsigned long long)count_instants(of.anchor_s, tlo, recs[f].first_event + 1) : 0ull;
        a.result[ch].first_event = recs[f].first_event;
        a.result[ch].n_dibits_after_first = sh_cnt[1023] - of.dibit_off - pre;
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
    uint8_t* dibits;            // [ch][dibit_stride]
    long dibit_stride;
    int64_t* sync_pos;          // nullable
    uint64_t* sync_dibit;       // nullable
    long sync_stride;
};

// K4: slice the anchored symbol instants of one tile.
__global__ __launch_bounds__(NT) void k_slice(SliceArgs a)
{
    __shared__ float BT[BT_N];
    __shared__ float THI[TB], TMID[TB], TLO[TB];
    __shared__ long shl[8];
    __shared__ int shi[8];
    const int tid = threadIdx.x, tile = blockIdx.x, ch = blockIdx.y;
    const long t0 = (long)tile * TB;
    const float* bbp = a.bb + (size_t)ch * a.bb_stride;
    const long tn = (a.n - t0 < TB) ? a.n - t0 : TB;
    const ScanOut co = a.outs[(size_t)ch * a.n_tiles + tile];

    for (int k = tid; k < BT_N; k += NT) BT[k] = bb_at(bbp, a.n_hist, a.n, t0 - HIST_BB + k);
    uint8_t evl[VPT];
    {
        const uint8_t* evp = a.events + (size_t)ch * a.ev_stride + t0;
#pragma unroll
        for (int u = 0; u < VPT; ++u) evl[u] = (tid * VPT + u < tn) ? evp[tid * VPT + u] : 0;
    }
    __syncthreads();

    long my_last = -1;
    int my_ev = 0;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid * VPT + u;
        if (evl[u]) {
            my_last = i;
            ++my_ev;
            float hi, mid, lo;
            sync_thresholds(BT, i - W + HIST_BB, hi, mid, lo);
            THI[i] = hi; TMID[i] = mid; TLO[i] = lo;
        }
    }
    const long incoming = block_excl_max(my_last, shl, tid);     // includes a barrier: thresholds visible
    // walk my 8 samples: count, then rank, then emit
    long cur = incoming;
    int cnt = 0;
    unsigned inst = 0;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid * VPT + u;
        if (i < tn) {
            bool is = false;
            if (cur >= 0) is = ((i - (cur - W)) % SPS) == 0;
            else if (co.valid) is = ((a.abs0 + t0 + i - co.anchor_s) % SPS) == 0;
            if (is) { ++cnt; inst |= 1u << u; }
            if (evl[u]) cur = i;
        }
    }
    int total;
    int rank = block_excl_sum(cnt, shi, tid, total);
    int evtotal;
    int evrank = block_excl_sum(my_ev, shi, tid, evtotal);
    uint8_t* out = a.dibits + (size_t)ch * a.dibit_stride + co.dibit_off;
    cur = incoming;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = tid * VPT + u;
        if (i < tn) {
            // the instant at index i (if any) is governed by events decided before i
            if ((inst >> u) & 1u) {
                float hi, mid, lo;
                if (cur >= 0) { hi = THI[cur]; mid = TMID[cur]; lo = TLO[cur]; }
                else { hi = co.hi; mid = co.mid; lo = co.lo; }
                const float v = BT[i + HIST_BB];
                out[rank++] = v >= hi ? 1 : v >= mid ? 0 : v >= lo ? 2 : 3;
            }
            if (evl[u]) {
                if (a.sync_pos && (long)(co.event_off + evrank) < a.sync_stride) {
                    a.sync_pos[(size_t)ch * a.sync_stride + co.event_off + evrank] = a.abs0 + t0 + i - W;
                    // index of the first dibit this detection governs = dibits for instants <= i
                    a.sync_dibit[(size_t)ch * a.sync_stride + co.event_off + evrank] = co.dibit_off + rank;
                }
                ++evrank;
                cur = i;
            }
        }
    }
}

}  // namespace p25k
