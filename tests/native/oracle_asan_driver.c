/* AddressSanitizer / UBSan driver for the CPU oracle (test infrastructure): every entry point on ragged chunks, both
 * symbol clocks, lock drops, the threaded throughput helper.  Built by `make -C oracle asan`; exit code 0 and no
 * sanitizer report = pass (tests/test_sanitizers.py). */
#include "../../oracle/p25fe_oracle.c"
#include "p25fe_spec.h"
#include <stdio.h>

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
    p25o_config c;
    memset(&c, 0, sizeof c);
    c.decim = P25FE_DECIM; c.t1 = P25FE_T1; c.t2 = P25FE_T2; c.boxcar = P25FE_BOXCAR; c.sps = P25FE_SPS; c.peak_w = P25FE_PEAK_W;
    c.sync_sign_mask = P25FE_SYNC_SIGN_MASK;
    memcpy(c.decim_taps, P25FE_DEFAULT_DECIM_TAPS, sizeof(float) * P25FE_T1);
    memcpy(c.chan_taps, P25FE_DEFAULT_CHAN_TAPS, sizeof(float) * P25FE_T2);
    memcpy(c.atan_c, P25FE_ATAN_COEFFS, sizeof(float) * P25FE_ATAN_NCOEF);
    c.fm_gain = P25FE_FM_GAIN; c.u8_scale = P25FE_U8_SCALE; c.u8_offset = P25FE_U8_OFFSET; c.boxcar_scale = P25FE_BOXCAR_SCALE; c.pi = P25FE_PI; c.half_pi = P25FE_HALF_PI;
    c.inv_npos = P25FE_SYNC_INV_NPOS; c.inv_nneg = P25FE_SYNC_INV_NNEG; c.rho2_n = P25FE_SYNC_RHO2_N; c.e_min = P25FE_SYNC_E_MIN;
    c.slice_frac = P25FE_SLICE_FRAC;
    c.clk_lookahead = P25FE_CLK_LOOKAHEAD; c.clk_tol_shift = P25FE_CLK_TOL_SHIFT; c.clk_dmax_log2 = P25FE_CLK_DMAX_LOG2;
    memcpy(c.clk_interp, P25FE_CLK_INTERP, sizeof c.clk_interp);

    enum { N = 200000 };
    unsigned seed = 7;
    uint8_t *u8 = malloc(2 * N);
    float *iq = malloc(sizeof(float) * 2 * N), *bb = malloc(sizeof(float) * (N / 5 + 8));
    double ph = 0.0;
    for (int i = 0; i < N; i++) {                                   /* a slow FM sweep plus noise: exercises sync candidates */
        ph += 0.02 * sin(i * 0.0013) + 0.03 * sin(i * 0.126);
        iq[2 * i] = 0.5f * (float)cos(ph) + 1e-3f * (float)(lcg(&seed) % 100 - 50);
        iq[2 * i + 1] = 0.5f * (float)sin(ph) + 1e-3f * (float)(lcg(&seed) % 100 - 50);
        u8[2 * i] = (uint8_t)(127.5f * (iq[2 * i] + 1.0f));
        u8[2 * i + 1] = (uint8_t)(127.5f * (iq[2 * i + 1] + 1.0f));
    }
    size_t total = 0;
    for (int mode = 0; mode < 2; mode++) {
        c.symbol_clock = mode;
        p25o_demod *d = p25o_demod_create(&c);
        p25o_recv *r = p25o_recv_create(&c);
        uint8_t *dib = malloc(N / 50 + 64);
        int64_t *sp = malloc(sizeof(int64_t) * 64);
        uint64_t *sd = malloc(sizeof(uint64_t) * 64);
        size_t off = 0, k = 0;
        while (off < N) {
            size_t n = 1 + lcg(&seed) % 9000;
            if (n > N - off) n = N - off;
            float pw;
            size_t nb = (k & 1) ? p25o_demod_u8(d, u8 + 2 * off, 2 * n, bb, &pw) : p25o_demod_cf32(d, iq + 2 * off, n, bb, (k & 2) ? &pw : NULL);
            size_t nd = 0, ns = 0;
            if (p25o_recv_feed(r, bb, nb, dib, N / 50 + 64, &nd, sp, sd, 64, &ns)) return 2;
            if (k % 7 == 3) p25o_recv_resync(r);
            total += nd + ns;
            off += n; k++;
        }
        int64_t t; int v; int64_t s; float thr[3]; uint64_t nd64;
        p25o_recv_state(r, &t, &v, &s, thr, &nd64);
        free(dib); free(sp); free(sd);
        p25o_recv_destroy(r);
        p25o_demod_destroy(d);
    }
    c.symbol_clock = 0;
    /* empty and tiny chunks */
    {
        p25o_demod *d = p25o_demod_create(&c);
        float pw;
        p25o_demod_cf32(d, iq, 0, bb, &pw);
        p25o_demod_u8(d, u8, 2, bb, NULL);
        p25o_demod_u8(d, u8, 8, bb, &pw);
        p25o_demod_destroy(d);
    }
    /* pre-decimator and channeliser */
    {
        p25o_predecim *p = p25o_predecim_create(P25FE_DEFAULT_PRE_TAPS, P25FE_T0, P25FE_PRE_DECIM);
        float *o = malloc(sizeof(float) * 2 * (N / 10 + 4));
        total += p25o_predecim_feed(p, iq, 5000, o) + p25o_predecim_feed(p, iq + 10000, 7, o) + p25o_predecim_feed(p, iq + 10014, 4000, o);
        p25o_predecim_destroy(p);
        float *cz = malloc(sizeof(float) * 2 * P25FE_CHZ_CHANNELS * 64);
        total += p25o_channelise(P25FE_DEFAULT_PRE_TAPS, P25FE_T0, P25FE_PRE_DECIM, P25FE_CHZ_CHANNELS, iq + 200, 100, 333, 12345, cz, 64);
        free(cz); free(o);
    }
    /* whole-path helpers */
    {
        uint8_t *dib = malloc(N / 50 + 64);
        total += (size_t)p25o_run_cf32(&c, iq, N, dib, N / 50 + 64);
        total += (size_t)p25o_run_cf32_mt(&c, iq, N, 3);
        uint64_t sd[2] = {0, 5};
        int64_t sp[2] = {1, 2};
        p25o_nid out[2];
        uint8_t dd[64];
        for (int i = 0; i < 64; i++) dd[i] = (uint8_t)(lcg(&seed) & 3);
        total += p25o_nid_decode((const uint64_t *)P25FE_NID_ROWS, P25FE_NID_T, dd, 36, sd, sp, 2, out);
        free(dib);
    }
    free(u8); free(iq); free(bb);
    printf("oracle sanitizer driver ok (%zu)\n", total);
    return 0;
}
