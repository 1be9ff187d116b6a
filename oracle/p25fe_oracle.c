/*
 * p25fe_oracle.c -- CPU restatement of the kchmck/p25rx IQ -> symbol hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product (p25rx_amd/, include/) never
 * includes, links or calls anything under oracle/.
 *
 * PARITY UNPINNED.  The reference keeps every number of this path (FIR taps, LUT values,
 * discriminator gain, sync/slicer rules) in crates that are not vendored in /root/reference
 * and cannot be fetched or compiled here (no rustc/cargo, no network):
 *     rtlsdr_iq 0.1.0, static_fir 0.2.0, demod_fm 1.0.0, moving_avg 0.1.0 (crates.io),
 *     static_decimate 1.0.0 @e9e00a0e, p25_filts 1.0.0 @0d34fc28, p25 1.0.0 @a96c564b (git)
 * (Cargo.lock:134-136, 320-322, 442-444, 458-460, 575-577, 655-657, 664-666), and none of
 * the reference's five #[test]s touches the path.  What this file follows from the reference
 * is therefore the STRUCTURE it can cite -- stage order, rates, buffer sizes, in-place
 * semantics, per-sample feed() objects whose state persists across chunks -- with the
 * build-defined numbers of docs/SPEC.md (frozen in tests/golden/spec.json) plugged in.
 * It is pinned instead against (i) an independent fp64 numpy/scipy model, (ii) physical
 * known-answer tests and (iii) the known symbols of a seeded C4FM modulator (tests/).
 *
 * Arithmetic contract (docs/SPEC.md section 2): fp32 throughout, every multiply-add that the
 * spec writes as fma() is one fused operation, nothing else may be contracted or
 * re-associated: build with -ffp-contract=off, no -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P25O_MAX_TAPS 64          /* config arrays */
#define P25O_FIR_CAP 128          /* capacity of one FIR object (the 80-tap pre-decimator is the longest) */
#define P25O_ATAN_NCOEF 8
#define P25O_SYNC_DIBITS 24

typedef struct { float re, im; } cf32;

/* Numbers of docs/SPEC.md; the tests fill this from tests/golden/spec.json. */
typedef struct {
    int32_t decim;                       /* src/demod.rs:50            -> 5  */
    int32_t t1, t2;                      /* tap counts                        */
    int32_t boxcar;                      /* src/demod.rs:52            -> 10 */
    int32_t sps;                         /* baseband samples per symbol -> 10 */
    int32_t peak_w;                      /* sync peak window (+-W)            */
    uint32_t sync_sign_mask;             /* bit j: sync symbol j (oldest first) is +3 */
    float decim_taps[P25O_MAX_TAPS];
    float chan_taps[P25O_MAX_TAPS];
    float atan_c[P25O_ATAN_NCOEF];
    float fm_gain, u8_scale, boxcar_scale;
    float pi, half_pi;
    float inv_npos, inv_nneg, rho2_n, e_min, slice_frac;
    /* SPEC 3.8b (tracking symbol clock; the reference's receiver has a fixed 10-sample stride, symbol_clock = 0) */
    int32_t symbol_clock;                /* 0: fixed stride (SPEC 3.8), 1: period tracked from sync word to sync word */
    int32_t clk_lookahead;               /* samples the receiver runs behind the baseband in mode 1 (2) */
    int32_t clk_tol_shift, clk_dmax_log2;
    float clk_interp[64 * 4];            /* cubic-Lagrange weights of the samples at -1, 0, +1, +2, row q: mu = q / 64 */
    /* rtlsdr_iq::IQ as a caller-supplied table (src/demod.rs:83): byte b -> u8_lut[b] when u8_lut_valid, else
     * fma(b, u8_scale, u8_offset) */
    float u8_offset;
    int32_t u8_lut_valid;
    float u8_lut[256];
    /* the last two constructor numbers of DemodTask::new as data (p25fe_config_t ABI 5):
     * Decimator::new(5) (src/demod.rs:50): output m comes from the input with absolute index 5 m + decim_phase (SPEC 3.2: 4);
     * MovingAverage::new(10) (src/demod.rs:52) as a table (SPEC 3.5): n_avg taps; all equal (avg_uniform) = a moving average,
     * summed newest first then scaled once -- `boxcar` / `boxcar_scale` above are that case's length and tap and stay in use
     * when n_avg == 0 --, otherwise a FIR: acc = +0, fma in tap order. */
    int32_t decim_phase;
    int32_t n_avg, avg_uniform;
    float avg_taps[P25O_MAX_TAPS];
} p25o_config;

/* ------------------------------------------------------------------------------------------
 * stage 1: rtlsdr_iq::IQ -- 65536-entry LUT indexed by the native-endian u16 made of two
 * consecutive bytes (src/demod.rs:74-76, 82-84).  Little endian: low byte = first byte = I.
 * Value (build-defined): fma((float)b, 2/255, -1); scale / offset or the whole 256-entry table can be configured (the
 * p25fe_config_t fields of the same names).
 * ---------------------------------------------------------------------------------------- */
static void build_iq_lut(cf32 *lut, const p25o_config *c)
{
    float v[256];
    for (uint32_t b = 0; b < 256u; b++)
        v[b] = c->u8_lut_valid ? c->u8_lut[b] : fmaf((float)b, c->u8_scale, c->u8_offset);
    for (uint32_t s = 0; s < 65536u; s++) {
        lut[s].re = v[s & 0xffu];
        lut[s].im = v[s >> 8];
    }
}

/* ------------------------------------------------------------------------------------------
 * static_fir::FirFilter<_>::feed(Complex32) -> Complex32 (src/demod.rs:29, 51, 93) and the
 * FIR inside static_decimate::Decimator (src/demod.rs:27, 50, 87).  History is a doubled
 * ring so the newest T samples are contiguous.  Accumulation order (SPEC 3.2): single
 * accumulator from +0, tap 0 (newest sample) to tap T-1 (oldest), one fma per component.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int ntaps, pos;
    float taps[P25O_FIR_CAP];
    cf32 hist[2 * P25O_FIR_CAP];
} fir_c;

static void fir_init(fir_c *f, const float *taps, int ntaps)
{
    memset(f, 0, sizeof *f);
    f->ntaps = ntaps;
    memcpy(f->taps, taps, sizeof(float) * (size_t)ntaps);
}

static inline void fir_push(fir_c *f, cf32 x)
{
    /* newest sample at hist[pos]; older ones at increasing addresses */
    f->pos = f->pos == 0 ? f->ntaps - 1 : f->pos - 1;
    f->hist[f->pos] = x;
    f->hist[f->pos + f->ntaps] = x;
}

static inline cf32 fir_eval(const fir_c *f)
{
    const cf32 *h = f->hist + f->pos;
    float re = 0.0f, im = 0.0f;
    for (int k = 0; k < f->ntaps; k++) {
        re = fmaf(f->taps[k], h[k].re, re);
        im = fmaf(f->taps[k], h[k].im, im);
    }
    return (cf32){re, im};
}

/* ------------------------------------------------------------------------------------------
 * demod_fm::FmDemod::feed (src/demod.rs:33, 54, 110): angle of s * conj(prev), times
 * fs / (2 pi dev).  atan2 is the spec's own polynomial (SPEC 3.4) so that two compilers
 * agree bit for bit; tests bound its distance to libm atan2f.
 * ---------------------------------------------------------------------------------------- */
static inline float spec_atan2f(const p25o_config *c, float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f)
        return 0.0f;
    float t = mn / mx;
    float s = t * t;
    float p = c->atan_c[P25O_ATAN_NCOEF - 1];
    for (int i = P25O_ATAN_NCOEF - 2; i >= 0; i--)
        p = fmaf(p, s, c->atan_c[i]);
    float r = p * t;
    if (ay > ax)
        r = c->half_pi - r;
    if (x < 0.0f)
        r = c->pi - r;
    if (y < 0.0f)
        r = -r;
    return r;
}

static inline float fm_feed(const p25o_config *c, cf32 *prev, cf32 s)
{
    float t = s.im * prev->im;
    float re = fmaf(s.re, prev->re, t);
    float u = s.re * prev->im;
    float im = fmaf(s.im, prev->re, -u);
    *prev = s;
    return spec_atan2f(c, im, re) * c->fm_gain;
}

/* ------------------------------------------------------------------------------------------
 * moving_avg::MovingAverage<f32>::feed (src/demod.rs:31, 52, 114).  SPEC 3.5: direct sum,
 * newest first, then one multiply by 1/10 -- no running sum, so it cannot drift and has
 * finite memory.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int len, pos, uniform; float taps[P25O_MAX_TAPS]; float hist[2 * P25O_MAX_TAPS]; } boxcar;

static inline float boxcar_feed(boxcar *b, float scale, float x)
{
    b->pos = b->pos == 0 ? b->len - 1 : b->pos - 1;
    b->hist[b->pos] = x;
    b->hist[b->pos + b->len] = x;
    const float *h = b->hist + b->pos;
    if (b->uniform) {                    /* MovingAverage: the reference's object */
        float acc = h[0];
        for (int i = 1; i < b->len; i++)
            acc = acc + h[i];
        return acc * scale;
    }
    float acc = 0.0f;                    /* any other table: a FIR like the two in front (SPEC 3.5, section 2) */
    for (int k = 0; k < b->len; k++)
        acc = fmaf(b->taps[k], h[k], acc);
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * demod::DemodTask (src/demod.rs:25-40): decimator, channel filter, moving average, FM
 * demodulator; state lives in the struct and persists across chunks (src/demod.rs:62-119).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    p25o_config cfg;
    cf32 *lut;
    fir_c decim;          /* Decimator<DecimFir>     src/demod.rs:27 */
    int decim_phase;      /* samples since last output; output when it reaches decim */
    fir_c bandpass;       /* FirFilter<BandpassFir>  src/demod.rs:29 */
    boxcar avg;           /* MovingAverage<f32>      src/demod.rs:31 */
    cf32 fm_prev;         /* FmDemod                 src/demod.rs:33 */
    cf32 *scratch;
    size_t scratch_cap;
} p25o_demod;

p25o_demod *p25o_demod_create(const p25o_config *cfg)
{
    if (cfg->t1 > P25O_MAX_TAPS || cfg->t2 > P25O_MAX_TAPS || cfg->boxcar > P25O_MAX_TAPS || cfg->n_avg < 0 || cfg->n_avg > P25O_MAX_TAPS ||
        cfg->decim_phase < 0 || cfg->decim_phase >= cfg->decim)
        return NULL;
    p25o_demod *d = calloc(1, sizeof *d);
    d->cfg = *cfg;
    d->lut = malloc(sizeof(cf32) * 65536);
    build_iq_lut(d->lut, cfg);
    fir_init(&d->decim, cfg->decim_taps, cfg->t1);       /* Decimator::new(5)  :50 */
    fir_init(&d->bandpass, cfg->chan_taps, cfg->t2);     /* FirFilter::new()   :51 */
    d->avg.len = cfg->boxcar;                            /* MovingAverage::new(10) :52 */
    d->avg.uniform = 1;
    if (cfg->n_avg > 0) {                                /* ... as a table */
        d->avg.len = cfg->n_avg;
        d->avg.uniform = cfg->avg_uniform;
        memcpy(d->avg.taps, cfg->avg_taps, sizeof(float) * (size_t)cfg->n_avg);
        if (cfg->avg_uniform) d->cfg.boxcar_scale = cfg->avg_taps[0];
    }
    /* Decimator: an output whenever the counter reaches `decim`; it starts so that the first one falls on input decim_phase */
    d->decim_phase = cfg->decim - 1 - cfg->decim_phase;
    return d;
}

void p25o_demod_destroy(p25o_demod *d)
{
    if (!d) return;
    free(d->lut);
    free(d->scratch);
    free(d);
}

/* static_decimate::Decimator::decim_in_place (src/demod.rs:87): filter + keep every 5th,
 * in place, return the new length.  Phase persists, so 16384-sample chunks give 3276 or
 * 3277 outputs (src/demod.rs:87-90).  SPEC 3.2: output m is produced by input n = 5m + decim_phase (4). */
static size_t decim_in_place(p25o_demod *d, cf32 *buf, size_t n)
{
    size_t len = 0;
    for (size_t i = 0; i < n; i++) {
        fir_push(&d->decim, buf[i]);
        if (++d->decim_phase == d->cfg.decim) {
            d->decim_phase = 0;
            buf[len++] = fir_eval(&d->decim);
        }
    }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * Stage 0 of BASELINE.json config 3 (2.4 Msps front end): 10:1 decimating FIR to 240 ksps in front of
 * DemodTask.  The reference has no such stage (SDR_SAMPLE_RATE is a fixed 240000, src/consts.rs:11); it is
 * built like static_decimate::Decimator (filter, keep every N-th, phase persists) with SPEC 3.0's taps:
 * output m is produced by input n = 10 m + 9, acc from +0, tap 0 on the newest sample.
 * ---------------------------------------------------------------------------------------- */
typedef struct { fir_c fir; int phase, decim; } p25o_predecim;

p25o_predecim *p25o_predecim_create(const float *taps, int ntaps, int decim)
{
    if (ntaps > P25O_FIR_CAP) return NULL;
    p25o_predecim *p = calloc(1, sizeof *p);
    fir_init(&p->fir, taps, ntaps);
    p->decim = decim;
    return p;
}

void p25o_predecim_destroy(p25o_predecim *p) { free(p); }

size_t p25o_predecim_feed(p25o_predecim *p, const float *iq, size_t n, float *out)
{
    const cf32 *x = (const cf32 *)iq;
    cf32 *y = (cf32 *)out;
    size_t len = 0;
    for (size_t i = 0; i < n; i++) {
        fir_push(&p->fir, x[i]);
        if (++p->phase == p->decim) {
            p->phase = 0;
            y[len++] = fir_eval(&p->fir);
        }
    }
    return len;
}

/* ------------------------------------------------------------------------------------------
 * Polyphase channeliser (SPEC 3.11; next row, SURVEY.md section 8f rank 4 -- no reference counterpart: the
 * reference tunes one channel, src/sdr.rs:64-65).  Stated the plain way -- mix channel c to DC, low-pass with SPEC
 * 3.0's taps, keep every 10th sample -- and evaluated in double precision:
 *   y_c[m] = sum_p h[p] x[n - p] e^{-j 2 pi c (n - p) / M},   n = absolute index of decimation instant m
 * (n = 9 mod 10, as in 3.0), x = 0 before the stream.  The GPU uses a factored DFT in fp32; parity is by tolerance.
 * x: owned sample 0 (n_hist valid samples before it), abs0: its absolute index.  out: [M][out_stride] cf32.
 * Returns the number of output instants. */
size_t p25o_channelise(const float *taps, int ntaps, int decim, int M, const float *iq, size_t n_hist, size_t n,
                       uint64_t abs0, float *out, size_t out_stride)
{
    const cf32 *x = (const cf32 *)iq;
    double *cs = malloc(sizeof(double) * 2 * (size_t)M);
    for (int k = 0; k < M; k++) {
        cs[2 * k] = cos(2.0 * 3.14159265358979323846 * k / M);
        cs[2 * k + 1] = sin(2.0 * 3.14159265358979323846 * k / M);
    }
    const size_t o0 = (size_t)((decim - 1 + decim - abs0 % (uint64_t)decim) % decim);
    size_t len = 0;
    for (size_t i = o0; i < n; i += (size_t)decim, len++) {
        for (int c = 0; c < M; c++) {
            double re = 0.0, im = 0.0;
            for (int p = 0; p < ntaps; p++) {
                const long j = (long)i - p;                       /* owned-relative index */
                if (j < -(long)n_hist) break;
                const uint64_t na = (uint64_t)((long)abs0 + j);    /* absolute index (j >= -n_hist >= -abs0) */
                const int k = (int)(((uint64_t)c * (na % (uint64_t)M)) % (uint64_t)M);
                const double wr = cs[2 * k], wi = -cs[2 * k + 1];  /* e^{-j 2 pi c n / M} */
                const double xr = x[j].re, xi = x[j].im, h = taps[p];
                re += h * (xr * wr - xi * wi);
                im += h * (xr * wi + xi * wr);
            }
            cf32 *y = (cf32 *)out + (size_t)c * out_stride + len;
            y->re = (float)re; y->im = (float)im;
        }
    }
    free(cs);
    return len;
}

/* demod::power_dbm (src/demod.rs:123-134), transcribed operation for operation. */
float p25o_power_dbm(const cf32 *samples, size_t n, float resistance)
{
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float a = samples[i].re * samples[i].re;
        float b = samples[i].im * samples[i].im;
        s = s + (a + b);                 /* s + x.norm_sqr()  :125-127 */
    }
    float avg = s / (float)n;
    float power = avg / resistance;      /* :130 */
    return 30.0f + 10.0f * log10f(power);/* :133 */
}

/* Stages after the LUT, on a chunk already in `samples` (src/demod.rs:86-114).
 * Returns the baseband length; *power (if non-NULL) gets power_dbm of the post-bandpass
 * chunk (src/demod.rs:97) -- the caller decides the every-4th-chunk throttle (:67, :95). */
static size_t demod_tail(p25o_demod *d, cf32 *samples, size_t n, float *bb, float *power)
{
    size_t len = decim_in_place(d, samples, n);                  /* :87-90  */
    for (size_t i = 0; i < len; i++) {                           /* :93     */
        fir_push(&d->bandpass, samples[i]);
        samples[i] = fir_eval(&d->bandpass);
    }
    if (power)
        *power = p25o_power_dbm(samples, len, 1.0f);             /* :97; an empty chunk folds to 0 / 0 = NaN, as the reference's does */
    for (size_t i = 0; i < len; i++)                             /* :109-111 */
        bb[i] = fm_feed(&d->cfg, &d->fm_prev, samples[i]);
    for (size_t i = 0; i < len; i++)                             /* :114    */
        bb[i] = boxcar_feed(&d->avg, d->cfg.boxcar_scale, bb[i]);
    return len;
}

static cf32 *scratch(p25o_demod *d, size_t n)
{
    if (n > d->scratch_cap) {
        free(d->scratch);
        d->scratch = malloc(sizeof(cf32) * n);
        d->scratch_cap = n;
    }
    return d->scratch;
}

/* One pass of the DemodTask::run loop body on one chunk of interleaved u8 I/Q
 * (src/demod.rs:70-117).  The reference's chunk is 32768 bytes (src/consts.rs:6); any even
 * size is accepted so tests can prove chunk-size invariance. */
size_t p25o_demod_u8(p25o_demod *d, const uint8_t *bytes, size_t nbytes, float *bb, float *power)
{
    size_t n = nbytes / 2;
    cf32 *samples = scratch(d, n);
    for (size_t i = 0; i < n; i++) {                             /* :74-84 */
        uint16_t s = (uint16_t)(bytes[2 * i] | (bytes[2 * i + 1] << 8));
        samples[i] = d->lut[s];
    }
    return demod_tail(d, samples, n, bb, power);
}

/* Same, for input that is already Complex32 (BASELINE.json configs 2-5). */
size_t p25o_demod_cf32(p25o_demod *d, const float *iq, size_t n, float *bb, float *power)
{
    cf32 *samples = scratch(d, n);
    if (n) memcpy(samples, iq, sizeof(cf32) * n);                /* (an empty chunk has no scratch: memcpy(NULL, ., 0) is undefined) */
    return demod_tail(d, samples, n, bb, power);
}

/* Export the post-bandpass complex samples too (for stage-level cross-checks). */
size_t p25o_demod_cf32_stages(p25o_demod *d, const float *iq, size_t n, float *chan_out, float *fm_out,
                              float *bb)
{
    cf32 *samples = scratch(d, n);
    if (n) memcpy(samples, iq, sizeof(cf32) * n);
    size_t len = decim_in_place(d, samples, n);
    for (size_t i = 0; i < len; i++) {
        fir_push(&d->bandpass, samples[i]);
        samples[i] = fir_eval(&d->bandpass);
    }
    if (chan_out)
        memcpy(chan_out, samples, sizeof(cf32) * len);
    for (size_t i = 0; i < len; i++) {
        float f = fm_feed(&d->cfg, &d->fm_prev, samples[i]);
        if (fm_out) fm_out[i] = f;
        bb[i] = boxcar_feed(&d->avg, d->cfg.boxcar_scale, f);
    }
    return len;
}

float p25o_atan2f(const p25o_config *c, float y, float x) { return spec_atan2f(c, y, x); }

/* ------------------------------------------------------------------------------------------
 * Symbol receiver: the front half of p25::message::receiver::MessageReceiver::feed(f32)
 * (src/recv.rs:207, src/replay.rs:44) down to the point where a dibit exists, and
 * MessageReceiver::resync (src/recv.rs:136, 179).  The p25 crate is not available, so the
 * rule is build-defined (SPEC 3.7-3.8):
 *   c[n]   = sum_j sign_j * b[n - 10 (23 - j)]          (frame-sync correlation, 24 taps)
 *   e[n]   = sum_j b[n - 10 (23 - j)]^2
 *   cand   = c > 0 && e >= e_min && c*c >= 24*0.85 * e   (normalised correlation >= ~0.92)
 *   detect = cand[n] && c[n] > c[n-i] && c[n] >= c[n+i], i = 1..W
 *   a detection at s sets thresholds from the sync word's own levels and anchors symbol
 *   instants n = s + 10 k (k >= 1); instant n is sliced under the latest detection s with
 *   s + W < n, so a re-anchor that moves the timing by up to half a symbol neither drops
 *   nor repeats a symbol.  A detection at s is decided when sample s + W has been fed;
 *   dibits have no lag.
 * ---------------------------------------------------------------------------------------- */
#define RING 512
typedef struct { int64_t d, n; int32_t usable, f; } p25o_clk;          /* a clock D / N; usable: it came from a sync-to-sync interval that passed 3.8b's test;
                                                                           f: pass 1 -- the sync position's fraction (quarter samples); table -- the anchor's offset in 1 / N samples */
typedef struct {
    p25o_config cfg;
    int64_t t;                  /* samples fed so far */
    float b[RING], c[RING];
    uint8_t cand[RING];
    int anchor_valid;
    int64_t anchor_s;
    float hi, mid, lo;
    /* symbol clock of the anchor in force: instants at anchor_s + j * per_d / per_n, j = 1, 2, ...  (10 / 1 in mode 0) */
    int64_t per_d, per_n, per_off, next_j, next_i;     /* per_off: SPEC 3.8c's fractional anchor (0 in modes 0 / 1): instants at anchor_s + (j per_d + per_off) / per_n */
    int next_q;
    int prev_valid;             /* a detection since the last lock drop: the next sync-to-sync interval is a period estimate */
    int64_t prev_s;
    int prev_f;                 /* ... and its position's fraction in quarter samples (SPEC 3.8b) */
    uint64_t n_dibits;
    /* SPEC 3.8c (symbol_clock = 2, resident ranges): the clock of detection k, counted from the creation of the receiver,
     * can be RECORDED (pass 1) and REPLACED (pass 2) -- see p25o_reslice_table */
    uint64_t n_det;
    p25o_clk *rec; size_t rec_cap;
    const p25o_clk *ovr; size_t ovr_len;
} p25o_recv;

p25o_recv *p25o_recv_create(const p25o_config *cfg)
{
    p25o_recv *r = calloc(1, sizeof *r);
    r->cfg = *cfg;
    r->per_d = cfg->sps; r->per_n = 1;
    return r;
}

void p25o_recv_destroy(p25o_recv *r) { free(r); }

/* SPEC 3.8c.  The streaming receiver gives a detection the period of the interval that ENDS at it, so the first frame of a lock
 * run (first detection, after a resync, after an implausible interval) is sliced at the nominal 10 / 1 and walks off the eye when the
 * sample clock is off.  A call that has the whole range in memory can do better: such a detection takes the period of the interval
 * that STARTS at it -- the backward clock of the NEXT detection -- if that one is usable.  Restated here as two passes of the
 * same receiver: pass 1 records (backward clock, usable) of every detection, p25o_reslice_table forms the table, pass 2 (a fresh
 * receiver fed the same samples and the same resyncs) takes its clocks from the table.  Detections do not depend on the clock,
 * so the k-th detection of both passes is the same sync word. */
void p25o_recv_record(p25o_recv *r, p25o_clk *rec, size_t cap) { r->rec = rec; r->rec_cap = cap; }
void p25o_recv_override(p25o_recv *r, const p25o_clk *tab, size_t len) { r->ovr = tab; r->ovr_len = len; }
uint64_t p25o_recv_n_det(const p25o_recv *r) { return r->n_det; }
void p25o_reslice_table(const p25o_clk *rec, size_t n, int sps, p25o_clk *out)
{
    for (size_t k = 0; k < n; k++) {
        if (rec[k].usable) out[k] = rec[k];
        else if (k + 1 < n && rec[k + 1].usable) { out[k] = rec[k + 1]; out[k].usable = 0; }
        else { out[k].d = sps; out[k].n = 1; out[k].usable = 0; }
        /* the fractional anchor: the instants start from s + f / 4 instead of s.  Clocks are D / (4 N_sym) (3.8b); the nominal 10 / 1
         * is written 40 / 4 so that the quarter fits; the offset is f N_sym = f n / 4 in units of 1 / n samples */
        if (out[k].n == 1) { out[k].d *= 4; out[k].n = 4; }
        out[k].f = (int32_t)(rec[k].f * (out[k].n / 4));
    }
}

/* MessageReceiver::resync (src/recv.rs:136, 179): drop lock (and with it the clock estimate). */
void p25o_recv_resync(p25o_recv *r) { r->anchor_valid = 0; r->prev_valid = 0; }

static inline float bget(const p25o_recv *r, int64_t n) { return n < 0 ? 0.0f : r->b[n & (RING - 1)]; }
static inline float cget(const p25o_recv *r, int64_t n) { return n < 0 ? 0.0f : r->c[n & (RING - 1)]; }

/* position of instant next_j of the current anchor: integer part and interpolation phase (SPEC 3.8b) */
static inline void clock_advance(p25o_recv *r)
{
    const int64_t num = r->next_j * r->per_d + r->per_off;      /* (> 0: |per_off| <= per_n / 2 < per_d) */
    r->next_i = r->anchor_s + num / r->per_n;
    r->next_q = (int)(((num % r->per_n) * 64) / r->per_n);
}

/* Feed n baseband samples (the `for &s in samples.iter()` loop of src/recv.rs:148-150).
 * Writes dibits (one per byte, values 0..3) and sync events (sample position of the sync
 * word's last symbol; index of the first dibit after it). Returns 0, or -1 on overflow.
 * Mode 1 (SPEC 3.8b) is the same loop run `clk_lookahead` samples behind the input: index u = t - L is processed when
 * sample t arrives, so that the 4-tap interpolation around an instant in [u, u + 1) has its two right-hand samples. */
int p25o_recv_feed(p25o_recv *r, const float *bb, size_t n, uint8_t *dibits, size_t cap, size_t *n_dibits,
                   int64_t *sync_pos, uint64_t *sync_dibit, size_t sync_cap, size_t *n_sync)
{
    const p25o_config *g = &r->cfg;
    const int W = g->peak_w, S = g->sps;
    const int track = g->symbol_clock != 0;
    const int L = track ? g->clk_lookahead : 0;
    size_t nd = 0, ns = 0;
    for (size_t i = 0; i < n; i++) {
        r->b[r->t & (RING - 1)] = bb[i];
        const int64_t t = r->t - L;               /* index processed now */
        r->t++;
        if (t < 0) continue;
        float c = 0.0f, e = 0.0f;
        for (int j = 0; j < P25O_SYNC_DIBITS; j++) {
            float v = bget(r, t - (int64_t)S * (P25O_SYNC_DIBITS - 1 - j));
            c = ((g->sync_sign_mask >> j) & 1u) ? c + v : c - v;
            e = fmaf(v, v, e);
        }
        r->c[t & (RING - 1)] = c;
        r->cand[t & (RING - 1)] = (c > 0.0f) && (e >= g->e_min) && (c * c >= g->rho2_n * e);

        /* 1. slice the instant whose integer position is t, under the latest detection s with s + W < t */
        if (r->anchor_valid && t == r->next_i) {
            float v;
            if (track) {
                const float *w = g->clk_interp + 4 * r->next_q;
                v = w[0] * bget(r, t - 1);
                v = fmaf(w[1], bget(r, t), v);
                v = fmaf(w[2], bget(r, t + 1), v);
                v = fmaf(w[3], bget(r, t + 2), v);
            } else {
                v = bget(r, t);
            }
            uint8_t d = v >= r->hi ? 1 : v >= r->mid ? 0 : v >= r->lo ? 2 : 3;
            if (nd >= cap) return -1;
            dibits[nd++] = d;
            r->n_dibits++;
            r->next_j++;
            clock_advance(r);
        }
        /* 2. decide detection at m = t - W (needs c[m-W .. m+W] = c[.. t]) */
        const int64_t m = t - W;
        if (m >= 0 && r->cand[m & (RING - 1)]) {
            float cm = cget(r, m);
            int peak = 1;
            for (int k = 1; k <= W; k++)
                peak = peak && (cm > cget(r, m - k)) && (cm >= cget(r, m + k));
            if (peak) {
                float P = 0.0f, N = 0.0f;
                for (int j = 0; j < P25O_SYNC_DIBITS; j++) {
                    float v = bget(r, m - (int64_t)S * (P25O_SYNC_DIBITS - 1 - j));
                    if ((g->sync_sign_mask >> j) & 1u) P = P + v; else N = N + v;
                }
                P = P * g->inv_npos;
                N = N * g->inv_nneg;
                float mid = (P + N) * 0.5f;
                float span = (P - N) * 0.5f;
                float dlt = span * g->slice_frac;
                r->mid = mid; r->hi = mid + dlt; r->lo = mid - dlt;
                /* SPEC 3.8b: the interval from the previous sync word, if lock was held throughout and it is a
                 * plausible whole number of symbols, is the period estimate D / N; otherwise the nominal 10 / 1.  The two
                 * positions enter with their fractions f (quarter samples: vertex of the parabola through the correlation
                 * peak and its neighbours) -- the period only; the anchor stays on the whole sample m */
                int fq = 0;
                {
                    const float cl = cget(r, m - 1), cr = cget(r, m + 1);
                    const float num = cl - cr, den = (cl - (cm + cm)) + cr;
                    if (den < 0.0f) {
                        const float q = (num * 0.5f) / den;
                        if (q == q) {
                            const float v = floorf(q * 4.0f + 0.5f);
                            fq = v < -2.0f ? -2 : v > 2.0f ? 2 : (int)v;
                        }
                    }
                }
                r->per_d = S; r->per_n = 1;
                int usable = 0;
                if (track && r->prev_valid) {
                    const int64_t dd = m - r->prev_s, nn = (dd + S / 2) / S;
                    const int64_t err = dd > S * nn ? dd - S * nn : S * nn - dd;
                    if (nn >= 1 && dd <= ((int64_t)1 << g->clk_dmax_log2) && (err << g->clk_tol_shift) <= S * nn) {
                        const int64_t d4 = 4 * dd + (fq - r->prev_f);
                        usable = 1;
                        if (d4 != 4 * S * nn) { r->per_d = d4; r->per_n = 4 * nn; }     /* (exactly nominal: written 10 / 1, the same instants) */
                    }
                }
                r->per_off = 0;
                if (r->rec && r->n_det < r->rec_cap) { p25o_clk k = {r->per_d, r->per_n, usable, fq}; r->rec[r->n_det] = k; }
                if (r->ovr && r->n_det < r->ovr_len) {                       /* SPEC 3.8c, pass 2 */
                    r->per_d = r->ovr[r->n_det].d; r->per_n = r->ovr[r->n_det].n; r->per_off = r->ovr[r->n_det].f;
                }
                r->n_det++;
                r->prev_s = m; r->prev_f = fq; r->prev_valid = 1;
                r->anchor_s = m;
                r->anchor_valid = 1;
                r->next_j = 1;
                clock_advance(r);
                if (ns < sync_cap) {
                    if (sync_pos) sync_pos[ns] = m;
                    if (sync_dibit) sync_dibit[ns] = r->n_dibits;
                }
                ns++;
            }
        }
    }
    *n_dibits = nd;
    if (n_sync) *n_sync = ns;
    return 0;
}

/* Introspection for tests (thresholds of the current anchor). */
void p25o_recv_state(const p25o_recv *r, int64_t *t, int *valid, int64_t *s, float *thr3, uint64_t *nd)
{
    *t = r->t; *valid = r->anchor_valid; *s = r->anchor_s;
    thr3[0] = r->hi; thr3[1] = r->mid; thr3[2] = r->lo; *nd = r->n_dibits;
}

/* ------------------------------------------------------------------------------------------
 * Whole-path convenience used by the cpu_baseline timing leg of bench.py: cf32 IQ in 16384-
 * sample chunks (the reference's BUF_SAMPLES, src/consts.rs:8) through DemodTask then the
 * receiver loop, like the two threads of src/main.rs:270-287 run back to back.
 * ---------------------------------------------------------------------------------------- */
int64_t p25o_run_cf32(const p25o_config *cfg, const float *iq, size_t n, uint8_t *dibits, size_t cap)
{
    enum { CH = 16384 };
    p25o_demod *d = p25o_demod_create(cfg);
    p25o_recv *r = p25o_recv_create(cfg);
    float *bb = malloc(sizeof(float) * CH);
    size_t total = 0;
    int64_t rc = 0;
    for (size_t off = 0; off < n; off += CH) {
        size_t m = n - off < CH ? n - off : CH;
        size_t len = p25o_demod_cf32(d, iq + 2 * off, m, bb, NULL);
        size_t nd = 0;
        if (p25o_recv_feed(r, bb, len, dibits + total, cap - total, &nd, NULL, NULL, 0, NULL)) { rc = -1; break; }
        total += nd;
    }
    free(bb);
    p25o_demod_destroy(d);
    p25o_recv_destroy(r);
    return rc ? rc : (int64_t)total;
}

/* The same port on several host cores (bench.py's second CPU figure, SURVEY.md section 8d ii): the capture is cut into
 * `nthreads` contiguous time shards, each run through p25o_run_cf32 from a fresh state on its own thread.  Symbols at
 * the shard boundaries are NOT stitched -- this is a throughput figure, not a parity path.  Returns the dibit total. */
#include <pthread.h>
typedef struct { const p25o_config *cfg; const float *iq; size_t n; int64_t out; } mt_job;
static void *mt_worker(void *p)
{
    mt_job *j = p;
    size_t cap = j->n / 50 + 64;
    uint8_t *dib = malloc(cap);
    j->out = dib ? p25o_run_cf32(j->cfg, j->iq, j->n, dib, cap) : -1;
    free(dib);
    return NULL;
}
int64_t p25o_run_cf32_mt(const p25o_config *cfg, const float *iq, size_t n, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    mt_job jobs[256];
    pthread_t th[256];
    int started = 0, failed = 0;
    for (int k = 0; k < nthreads; k++) {
        size_t lo = n * (size_t)k / (size_t)nthreads, hi = n * (size_t)(k + 1) / (size_t)nthreads;
        jobs[k].cfg = cfg; jobs[k].iq = iq + 2 * lo; jobs[k].n = hi - lo; jobs[k].out = 0;
        if (pthread_create(&th[k], NULL, mt_worker, &jobs[k])) { failed = 1; break; }
        started++;
    }
    int64_t total = failed ? -1 : 0;
    for (int k = 0; k < started; k++) {      /* the workers use jobs[] on this stack: always join what was started */
        pthread_join(th[k], NULL);
        total = (total < 0 || jobs[k].out < 0) ? -1 : total + jobs[k].out;
    }
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Network identifier after the frame sync (what MessageReceiver reports as
 * MessageEvent::PacketNID, src/recv.rs:216-222, and policy.handle_nid consumes, src/policy.rs:92).
 * TIA-102.BAAA: 64 bits = BCH(63,16,23) code word (NAC 12 bits, DUID 4 bits, 47 parity bits) + 1 extra bit,
 * sent as 32 dibits right after the 24-dibit frame sync with one status symbol interleaved at dibit 35
 * counted from the start of the frame sync (= 11 dibits into the NID).  Decoding (SPEC 3.9): the code word
 * nearest in Hamming distance among all 65536 (smallest data word on ties), valid iff distance <= t = 11 --
 * the same result as a bounded-distance BCH decoder.  `rows` are the generator-matrix rows of p25fe_spec.h.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t raw;        /* the 64 received bits, first transmitted bit in bit 63 */
    int64_t sync_pos;    /* copied from the sync event */
    uint16_t nac;
    uint8_t duid;
    uint8_t n_errors;    /* Hamming distance to the chosen code word (saturated at 255) */
    int32_t valid;       /* 1: decoded, 0: more than t errors, -1: the dibit stream ends inside the NID */
} p25o_nid;

size_t p25o_nid_decode(const uint64_t *rows, int t, const uint8_t *dibits, size_t n_dibits, const uint64_t *sync_dibit,
                       const int64_t *sync_pos, size_t n_sync, p25o_nid *out)
{
    uint64_t lo[256], hi[256];                 /* code word of data = (h << 8) | l is hi[h] ^ lo[l] */
    for (int v = 0; v < 256; v++) {
        uint64_t a = 0, b = 0;
        for (int i = 0; i < 8; i++)
            if (v & (1 << i)) { a ^= rows[i]; b ^= rows[8 + i]; }
        lo[v] = a; hi[v] = b;
    }
    for (size_t k = 0; k < n_sync; k++) {
        p25o_nid r;
        memset(&r, 0, sizeof r);
        r.sync_pos = sync_pos ? sync_pos[k] : 0;
        const uint64_t D = sync_dibit[k];
        if (D + 33 > n_dibits) { r.valid = -1; out[k] = r; continue; }
        uint64_t raw = 0;
        for (int j = 0; j < 33; j++) {
            if (j == 11) continue;             /* status symbol */
            raw = (raw << 2) | (uint64_t)(dibits[D + j] & 3u);
        }
        r.raw = raw;
        const uint64_t cw = raw >> 1;
        int best = 64; unsigned bd = 0;
        for (unsigned d = 0; d < 65536u; d++) {
            const int dist = __builtin_popcountll((hi[d >> 8] ^ lo[d & 255u]) ^ cw);
            if (dist < best) { best = dist; bd = d; }
        }
        r.nac = (uint16_t)(bd >> 4);
        r.duid = (uint8_t)(bd & 15u);
        r.n_errors = (uint8_t)(best > 255 ? 255 : best);
        r.valid = best <= t ? 1 : 0;
        out[k] = r;
    }
    return n_sync;
}

int p25o_has_fma(void) { return __builtin_cpu_supports("fma"); }
