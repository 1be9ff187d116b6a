#!/usr/bin/env python3
"""Generate the build-defined numeric spec of the p25fe hot path.

The reference (kchmck/p25rx) keeps every coefficient of the IQ->symbol path in
un-vendored crates (p25_filts, demod_fm, rtlsdr_iq, p25 -- see SURVEY.md section 0), so
this build owns its numbers.  This script derives them in fp64 from closed-form
definitions, rounds to fp32 once, and freezes them as hex-float literals in

  * include/p25fe_spec.h              (C header: default config of the C-ABI library)
  * tests/golden/spec.json            (same values as data, for the oracle-side tests)

It is a design-time tool: nothing imports it at run time and numpy's i0/kaiser are
not needed on the GPU box.  Re-running it must reproduce the committed files bit for
bit (tests/test_spec.py checks that).

Definitions
-----------
decimator  : T1 = 31 taps, Kaiser(beta=7.0)-windowed sinc, cutoff 12 kHz at fs = 240 kHz,
             unity DC gain.  Anti-alias for the 5:1 rate change of src/demod.rs:50.
channel    : T2 = 41 taps, Kaiser(beta=5.0)-windowed sinc, cutoff 6.25 kHz at fs = 48 kHz,
             unity DC gain.  "Channel-select lowpass" of src/demod.rs:28-29.
pre-decim  : T0 = 80 taps, Kaiser(beta=7.0)-windowed sinc, cutoff 60 kHz at fs = 2.4 MHz, unity DC
             gain: the 10:1 stage of BASELINE.json config 3 (2.4 Msps front end; no reference counterpart).
atan       : odd polynomial t*(c0 + c1 t^2 + ... + c7 t^14) ~ atan(t) on [0, 1], weighted
             least squares on 4096 Chebyshev nodes, coefficients rounded to fp32.
fm gain    : fs / (2 pi dev) with fs = 48000, dev = 5000 (src/demod.rs:54).
nid        : network identifier that follows the frame sync: 16 data bits (NAC 12, DUID 4) in a BCH(63,16,23)
             code word + 1 extra bit; generator polynomial 6331141367235453 (octal, TIA-102.BAAA).  This script
             verifies that g(x) has degree 47, divides x^63 + 1 and gives minimum weight 23 (it refuses otherwise).
sync       : P25 frame sync word 0x5575F5FF77FF (TIA-102.BAAA), dibit 01 -> +3, 11 -> -3.
clock      : tracking symbol clock (SPEC 3.8b): 64 x 4 table of cubic-Lagrange interpolation weights for the samples at
             -1, 0, +1, +2 around a fractional instant mu = q / 64 (row 0 is exactly 0, 1, 0, 0, so that a period of exactly
             10 samples reproduces the fixed-stride receiver bit for bit); lookahead 2 samples; a sync-to-sync interval
             D = 10 N + r is accepted as a period estimate D / N when |r| * 1024 <= 10 N and D <= 2^24.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T1, T2 = 31, 41
T0, PRE_DECIM, FS_WIDE = 80, 10, 2400000.0
CHZ_M = 192                          # channeliser: 2.4 Msps / 12.5 kHz raster (SPEC 3.11)
FS_IN, FS_BB = 240000.0, 48000.0
DECIM = 5
SPS = 10                      # baseband samples per symbol (48000 / 4800)
SYNC_WORD = 0x5575F5FF77FF    # 48 bits, 24 dibits
SYNC_DIBITS = 24
NID_GEN_POLY = int("6331141367235453", 8)
CLK_PHASES, CLK_TAPS, CLK_LOOKAHEAD, CLK_TOL_SHIFT, CLK_DMAX_LOG2 = 64, 4, 2, 10, 24


def interp_table():
    rows = []
    for q in range(CLK_PHASES):
        mu = q / float(CLK_PHASES)
        rows.append([-mu * (mu - 1.0) * (mu - 2.0) / 6.0, (mu + 1.0) * (mu - 1.0) * (mu - 2.0) / 2.0,
                     -(mu + 1.0) * mu * (mu - 2.0) / 2.0, (mu + 1.0) * mu * (mu - 1.0) / 6.0])
    t = np.array(rows, dtype=np.float64)
    assert np.allclose(t.sum(axis=1), 1.0) and list(t[0]) == [0.0, 1.0, 0.0, 0.0]
    return t.astype(np.float32)


def kaiser_sinc(ntaps, cutoff_hz, fs, beta):
    n = np.arange(ntaps, dtype=np.float64) - (ntaps - 1) / 2.0
    fc = cutoff_hz / fs
    h = 2.0 * fc * np.sinc(2.0 * fc * n) * np.kaiser(ntaps, beta)
    h = h / h.sum()
    # enforce exact symmetry before rounding
    h = 0.5 * (h + h[::-1])
    return h.astype(np.float32)


def atan_coeffs(ncoef=8):
    k = np.arange(4096, dtype=np.float64)
    t = 0.5 * (1.0 - np.cos(np.pi * (k + 0.5) / 4096.0))      # Chebyshev nodes on [0,1]
    t = np.concatenate([t, [1.0]])
    target = np.arctan(t)
    A = np.stack([t ** (2 * i + 1) for i in range(ncoef)], axis=1)
    # iteratively reweighted least squares -> close to minimax
    w = np.ones_like(t)
    for _ in range(60):
        c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        err = np.abs(A @ c - target)
        w = w * (1.0 + 4.0 * err / err.max())
        w /= w.mean()
    return c.astype(np.float32)


def sync_symbols():
    syms = []
    for i in range(SYNC_DIBITS):
        d = (SYNC_WORD >> (2 * (SYNC_DIBITS - 1 - i))) & 3
        syms.append({0b01: 3, 0b11: -3, 0b00: 1, 0b10: -1}[d])
    return syms


def gf2_mod(a, b):
    db = b.bit_length()
    while a.bit_length() >= db:
        a ^= b << (a.bit_length() - db)
    return a


def nid_encode(data16):
    """Systematic BCH(63,16): 16 data bits, then the remainder of data * x^47 modulo g(x)."""
    m = data16 << 47
    return m | gf2_mod(m, NID_GEN_POLY)


def check_nid_code():
    assert NID_GEN_POLY.bit_length() - 1 == 47
    assert gf2_mod((1 << 63) | 1, NID_GEN_POLY) == 0
    rows = [nid_encode(1 << i) for i in range(16)]
    cw, prev, minw = 0, 0, 64
    for i in range(1, 1 << 16):
        gray = i ^ (i >> 1)
        cw ^= rows[(gray ^ prev).bit_length() - 1]
        prev = gray
        minw = min(minw, bin(cw).count("1"))
    assert minw == 23, minw
    return rows


def rc_taps(alpha=0.2, span_symbols=4, sps=SPS):
    """Raised-cosine pulse (roll-off alpha) over span_symbols symbols at sps samples per symbol, unity DC gain: the NON-default
    example of a post-discriminator filter (p25fe_config_t.avg_taps) -- the 'raised-cosine matched filter' BASELINE.json's
    north_star names.  (The reference's own receive filter is the 10-sample moving average: with C4FM's transmit shaping
    that integrate-and-dump filter is what makes the cascade Nyquist; this table is a configuration example, not a better
    receiver.)"""
    n = span_symbols * sps + 1
    t = (np.arange(n, dtype=np.float64) - (n - 1) / 2.0) / sps
    den = 1.0 - (2.0 * alpha * t) ** 2
    h = np.where(np.abs(den) < 1e-12, np.pi / 4.0 * np.sinc(1.0 / (2.0 * alpha)), np.sinc(t) * np.cos(np.pi * alpha * t) / np.where(np.abs(den) < 1e-12, 1.0, den))
    h = h / h.sum()
    h = 0.5 * (h + h[::-1])
    return h.astype(np.float32)


def hexf(x):
    return float(np.float32(x)).hex() + "f"


def main():
    dec = kaiser_sinc(T1, 12000.0, FS_IN, 7.0)
    chan = kaiser_sinc(T2, 6250.0, FS_BB, 5.0)
    pre = kaiser_sinc(T0, 60000.0, FS_WIDE, 7.0)
    atc = atan_coeffs()
    fm_gain = np.float32(FS_BB / (2.0 * np.pi * 5000.0))
    syms = sync_symbols()
    assert all(abs(s) == 3 for s in syms)
    nid_rows = check_nid_code()
    sync_sign_mask = 0
    for j, s in enumerate(syms):           # bit j set <=> symbol j (oldest first) is +3
        if s > 0:
            sync_sign_mask |= 1 << j
    npos = sum(1 for s in syms if s > 0)
    nneg = SYNC_DIBITS - npos

    spec = {
        "decim": DECIM, "sps": SPS, "t1": T1, "t2": T2,
        "decim_taps": [float(x) for x in dec],
        "chan_taps": [float(x) for x in chan],
        "pre_decim": PRE_DECIM, "t0": T0, "pre_taps": [float(x) for x in pre],
        "atan_coeffs": [float(x) for x in atc],
        "fm_gain": float(fm_gain),
        "u8_scale": float(np.float32(2.0 / 255.0)), "u8_offset": -1.0,
        "fm_deviation_hz": 5000, "fm_sample_rate_hz": int(FS_BB),
        "boxcar_len": 10, "boxcar_scale": float(np.float32(0.1)),
        # the post-discriminator filter as a TABLE (p25fe_config_t.avg_taps, ABI 5): ten equal taps = MovingAverage::new(10)
        "avg_taps": [float(np.float32(0.1))] * 10, "decim_phase": 4,
        "rc_avg_taps": [float(x) for x in rc_taps()],
        "sync_symbols": syms, "sync_sign_mask": sync_sign_mask,
        "sync_npos": npos, "sync_nneg": nneg,
        "sync_inv_npos": float(np.float32(1.0 / npos)),
        "sync_inv_nneg": float(np.float32(1.0 / nneg)),
        "sync_rho2_n": float(np.float32(SYNC_DIBITS * 0.85)),
        "sync_e_min": float(np.float32(SYNC_DIBITS * 0.05 * 0.05)),
        "sync_peak_w": 5,
        "slice_frac": float(np.float32(2.0 / 3.0)),
        "pi": float(np.float32(np.pi)), "half_pi": float(np.float32(np.pi / 2.0)),
        "chz_channels": CHZ_M, "chz_spacing_hz": FS_WIDE / CHZ_M,
        "chz_w": [[float(np.float32(np.round(np.cos(2 * np.pi * k / CHZ_M), 15))),     # quarter turns exactly 0 / +-1
                   float(np.float32(np.round(np.sin(2 * np.pi * k / CHZ_M), 15)))] for k in range(CHZ_M)],
        "nid_gen_poly": NID_GEN_POLY, "nid_rows": nid_rows, "nid_t": 11, "nid_status_pos": 35,
        "clk_phases": CLK_PHASES, "clk_lookahead": CLK_LOOKAHEAD, "clk_tol_shift": CLK_TOL_SHIFT, "clk_dmax_log2": CLK_DMAX_LOG2,
        "clk_interp": [[float(v) for v in row] for row in interp_table()],
    }

    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "golden", "spec.json"), "w") as f:
        json.dump(spec, f, indent=1, sort_keys=True)
        f.write("\n")

    def arr(name, vals):
        body = ",\n    ".join(", ".join(hexf(v) for v in vals[i:i + 4]) for i in range(0, len(vals), 4))
        return "static P25FE_SPEC_CONST float %s[%d] = {\n    %s\n};\n" % (name, len(vals), body)

    h = []
    h.append("/* GENERATED by tools/gen_spec.py -- do not edit.  Build-defined numeric spec of the\n"
             " * p25fe hot path (the reference keeps these numbers in un-vendored crates: p25_filts,\n"
             " * demod_fm, rtlsdr_iq, p25 -- SURVEY.md section 0).  fp64-derived, rounded once to fp32. */\n")
    h.append("#ifndef P25FE_SPEC_H\n#define P25FE_SPEC_H\n\n")
    h.append("#ifdef __cplusplus\n#define P25FE_SPEC_CONST constexpr   /* usable in constant expressions (twiddle tables) */\n"
             "#else\n#define P25FE_SPEC_CONST const\n#endif\n\n")
    h.append("#define P25FE_DECIM %d            /* src/demod.rs:50 */\n" % DECIM)
    h.append("#define P25FE_SPS %d               /* 48000 / 4800 baud; src/demod.rs:52 */\n" % SPS)
    h.append("#define P25FE_T1 %d               /* decimator taps */\n" % T1)
    h.append("#define P25FE_T2 %d               /* channel filter taps */\n" % T2)
    h.append("#define P25FE_BOXCAR %d           /* src/demod.rs:52 */\n" % 10)
    h.append("#define P25FE_PRE_DECIM %d        /* 2.4 Msps -> 240 ksps (BASELINE.json config 3) */\n" % PRE_DECIM)
    h.append("#define P25FE_T0 %d               /* pre-decimator taps */\n" % T0)
    h.append("#define P25FE_ATAN_NCOEF %d\n" % len(atc))
    h.append("#define P25FE_SYNC_DIBITS %d\n" % SYNC_DIBITS)
    h.append("#define P25FE_SYNC_SPAN %d        /* (24-1)*10 baseband samples */\n" % ((SYNC_DIBITS - 1) * SPS))
    h.append("#define P25FE_SYNC_SIGN_MASK 0x%06xu /* bit j: sync symbol j (oldest first) is +3 */\n" % sync_sign_mask)
    h.append("#define P25FE_SYNC_NPOS %d\n#define P25FE_SYNC_NNEG %d\n" % (npos, nneg))
    h.append("#define P25FE_SYNC_INV_NPOS %s\n" % hexf(spec["sync_inv_npos"]))
    h.append("#define P25FE_SYNC_INV_NNEG %s\n" % hexf(spec["sync_inv_nneg"]))
    h.append("#define P25FE_SYNC_RHO2_N %s   /* 24 * 0.85 */\n" % hexf(spec["sync_rho2_n"]))
    h.append("#define P25FE_SYNC_E_MIN %s    /* 24 * 0.05^2 */\n" % hexf(spec["sync_e_min"]))
    h.append("#define P25FE_PEAK_W %d\n" % 5)
    h.append("#define P25FE_SLICE_FRAC %s    /* 2/3 */\n" % hexf(spec["slice_frac"]))
    h.append("#define P25FE_FM_GAIN %s       /* 48000 / (2 pi 5000); src/demod.rs:54 */\n" % hexf(fm_gain))
    h.append("#define P25FE_U8_SCALE %s      /* 2/255 */\n" % hexf(spec["u8_scale"]))
    h.append("#define P25FE_U8_OFFSET %s\n" % hexf(spec["u8_offset"]))
    h.append("#define P25FE_FM_DEVIATION_HZ %d       /* src/demod.rs:54 */\n#define P25FE_FM_SAMPLE_RATE_HZ %d   /* src/consts.rs:13 */\n"
             % (spec["fm_deviation_hz"], spec["fm_sample_rate_hz"]))
    h.append("#define P25FE_BOXCAR_SCALE %s  /* 1/10 */\n" % hexf(spec["boxcar_scale"]))
    h.append("#define P25FE_DECIM_PHASE %d     /* output m of the 5:1 decimator is produced by input 5 m + 4 (docs/SPEC.md 3.2) */\n" % spec["decim_phase"])
    h.append("#define P25FE_PI %s\n#define P25FE_HALF_PI %s\n" % (hexf(spec["pi"]), hexf(spec["half_pi"])))
    h.append("#define P25FE_NID_GEN_POLY 0x%xULL   /* octal 6331141367235453: BCH(63,16,23), TIA-102.BAAA */\n" % NID_GEN_POLY)
    h.append("#define P25FE_NID_T 11                /* correctable bit errors */\n")
    h.append("#define P25FE_NID_DIBITS 32           /* 64 bits = 63-bit code word + 1 extra bit */\n")
    h.append("#define P25FE_NID_STATUS_POS 35       /* dibit index (from the start of the frame sync) of the status symbol inside the NID */\n\n")
    h.append("/* rows of the systematic generator matrix: code word of data bit i (bit 62 = first transmitted bit) */\n")
    h.append("static const unsigned long long P25FE_NID_ROWS[16] = {\n    " + ",\n    ".join("0x%016xULL" % r for r in nid_rows) + "\n};\n\n")
    h.append(arr("P25FE_DEFAULT_DECIM_TAPS", list(dec)))
    h.append("\n")
    h.append(arr("P25FE_DEFAULT_CHAN_TAPS", list(chan)))
    h.append("\n")
    h.append(arr("P25FE_DEFAULT_PRE_TAPS", list(pre)))
    h.append("\n/* post-discriminator filter (docs/SPEC.md 3.5): MovingAverage::new(10) as a table of ten equal taps */\n")
    h.append(arr("P25FE_DEFAULT_AVG_TAPS", spec["avg_taps"]))
    h.append("\n/* a NON-default example for p25fe_config_t.avg_taps: raised cosine, roll-off 0.2, 4 symbols at 10 samples per symbol */\n")
    h.append("#define P25FE_RC_AVG_N %d\n" % len(spec["rc_avg_taps"]))
    h.append(arr("P25FE_RC_AVG_TAPS", spec["rc_avg_taps"]))
    h.append("\n")
    h.append(arr("P25FE_ATAN_COEFFS", list(atc)))
    h.append("\n/* channeliser (SPEC 3.11): e^{+j 2 pi k / 192} as (cos, sin) pairs, k = 0..191 */\n")
    h.append("#define P25FE_CHZ_CHANNELS %d       /* 2.4 Msps / 12.5 kHz raster */\n" % CHZ_M)
    h.append(arr("P25FE_CHZ_W", [v for pair in spec["chz_w"] for v in pair]))
    h.append("\n/* tracking symbol clock (SPEC 3.8b): cubic-Lagrange weights of the samples at -1, 0, +1, +2 for mu = q / 64 */\n")
    h.append("#define P25FE_CLK_PHASES %d\n#define P25FE_CLK_LOOKAHEAD %d     /* samples the receiver runs behind the baseband */\n" % (CLK_PHASES, CLK_LOOKAHEAD))
    h.append("#define P25FE_CLK_TOL_SHIFT %d    /* interval 10 N + r accepted as a period when |r| << shift <= 10 N */\n" % CLK_TOL_SHIFT)
    h.append("#define P25FE_CLK_DMAX_LOG2 %d    /* ... and the interval is at most 2^24 samples */\n" % CLK_DMAX_LOG2)
    h.append(arr("P25FE_CLK_INTERP", [v for row in interp_table() for v in row]))
    h.append("\n#endif /* P25FE_SPEC_H */\n")
    os.makedirs(os.path.join(ROOT, "include"), exist_ok=True)
    with open(os.path.join(ROOT, "include", "p25fe_spec.h"), "w") as f:
        f.write("".join(h))

    # report
    def resp(hh, fs, freqs):
        n = np.arange(len(hh))
        return [20 * np.log10(abs(np.sum(hh.astype(np.float64) * np.exp(-2j * np.pi * f / fs * n))) + 1e-300)
                for f in freqs]
    print("decim taps sum", float(dec.astype(np.float64).sum()))
    print("decim resp dB @ 0,6.25k,12k,24k,41k,48k:", np.round(resp(dec, FS_IN, [0, 6250, 12000, 24000, 41000, 48000]), 2))
    print("pre   resp dB @ 0,12k,24k,41k,120k,199k,240k:", np.round(resp(pre, FS_WIDE, [0, 12000, 24000, 41000, 120000, 199000, 240000]), 2))
    print("chan  resp dB @ 0,3k,5k,6.25k,9k,12.5k:", np.round(resp(chan, FS_BB, [0, 3000, 5000, 6250, 9000, 12500]), 2))
    t = np.linspace(0, 1, 200001)
    p = np.zeros_like(t)
    for c in atc[::-1].astype(np.float64):
        p = p * t * t + c
    p = p * t
    print("atan poly max abs err (fp64 eval of fp32 coeffs):", np.abs(p - np.arctan(t)).max())
    print("sync symbols", syms, "npos", npos, "nneg", nneg, "mask", hex(sync_sign_mask))


if __name__ == "__main__":
    sys.exit(main())
