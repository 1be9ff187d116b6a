"""An INDEPENDENT restatement of docs/SPEC.md 3.7 + 3.8 (frame-sync correlation, peak test, thresholds, fixed-stride symbol
clock, 4-level slicer, resync) written from the SPEC's text as a BATCH computation over a whole baseband array in numpy
float32 -- no per-sample state machine, no chunking, nothing shared with oracle/p25fe_oracle.c, which is a streaming
`feed()` object like the reference's.  tests/test_oracle.py checks that the two agree dibit for dibit: a second pair of
eyes on the oracle's receiver, which is the parity target of the HIP path (test infrastructure).

Arithmetic: every SPEC operation on float32 arrays in the SPEC's order; the one fused multiply-add of the section
(e[n] = fma(v, v, acc)) is evaluated in float64 and rounded once (v * v is exact in float64; the sum then rounds twice,
which can differ from a true fma in the last bit -- the energy only enters two threshold comparisons, and
`margins()` reports how close any candidate came to them so that a test can tell a razor's-edge case from a defect)."""
import numpy as np

F = np.float32


def _shifted(b, k):
    """b[n - k] for every n, zero where n - k < 0"""
    if k == 0:
        return b
    out = np.zeros_like(b)
    out[k:] = b[:-k] if k < len(b) else 0
    return out


class Model:
    def __init__(self, spec):
        self.mask = int(spec["sync_sign_mask"])
        self.sps = int(spec["sps"])
        self.W = int(spec["sync_peak_w"])
        self.rho2_n = F(spec["sync_rho2_n"])
        self.e_min = F(spec["sync_e_min"])
        self.inv_p, self.inv_n = F(spec["sync_inv_npos"]), F(spec["sync_inv_nneg"])
        self.frac = F(spec["slice_frac"])
        self.signs = [(self.mask >> j) & 1 for j in range(24)]        # j = 0: oldest symbol of the sync word

    def correlate(self, b):
        """c[n], e[n] of SPEC 3.7 for every n (b[n] = 0 for n < 0)"""
        b = np.ascontiguousarray(b, dtype=F)
        c = np.zeros(len(b), dtype=F)
        e = np.zeros(len(b), dtype=F)
        for j in range(24):
            v = _shifted(b, self.sps * (23 - j))
            c = (c + v) if self.signs[j] else (c - v)
            e = (v.astype(np.float64) * v.astype(np.float64) + e.astype(np.float64)).astype(F)
        return c, e

    def detections(self, b):
        """sync positions s (ascending) with their thresholds (hi, mid, lo); a detection needs sample s + W to exist"""
        b = np.ascontiguousarray(b, dtype=F)
        n = len(b)
        c, e = self.correlate(b)
        cand = (c > 0) & (e >= self.e_min) & (c * c >= self.rho2_n * e)
        det = cand.copy()
        for i in range(1, self.W + 1):
            left = _shifted(c, i)                                     # c[s - i] (zero in front of the stream: the correlation of zeros)
            right = np.full(n, np.inf, dtype=F)                       # c[s + i]: not there yet -> no decision
            right[:n - i] = c[i:] if i < n else 0
            det &= (c > left) & (c >= right)
        pos = np.nonzero(det)[0]
        out = []
        for s in pos:
            P = F(0.0)
            N = F(0.0)
            for j in range(24):
                idx = s - self.sps * (23 - j)
                v = b[idx] if idx >= 0 else F(0.0)
                if self.signs[j]:
                    P = F(P + v)
                else:
                    N = F(N + v)
            P = F(P * self.inv_p)
            N = F(N * self.inv_n)
            mid = F(F(P + N) * F(0.5))
            span = F(F(P - N) * F(0.5))
            d = F(span * self.frac)
            out.append((int(s), F(mid + d), mid, F(mid - d)))
        self._c, self._e, self._cand = c, e, cand
        return out

    def margins(self):
        """smallest relative distance of any position with c > 0 to the two energy thresholds of the last detections() call"""
        c, e = self._c.astype(np.float64), self._e.astype(np.float64)
        m = c > 0
        if not m.any():
            return np.inf
        a = np.abs(e[m] - float(self.e_min)) / float(self.e_min)
        r = np.abs(c[m] * c[m] - float(self.rho2_n) * e[m]) / np.maximum(c[m] * c[m], 1e-30)
        return float(min(a.min(), r.min()))

    def receive_tracking(self, b, spec, resync=(), reslice=False):
        """SPEC 3.8b (symbol_clock = 1): every detection carries a clock D / N measured from the previous sync word, instants
        s + (j D) div N are read by the 4-tap interpolation of their phase; the receiver runs L samples behind the
        baseband (it processes index u when sample u + L exists), a drop before sample q removes instants >= q - L.
        reslice (SPEC 3.8c, symbol_clock = 2, one resident range): a detection whose own interval gave no clock takes the clock of
        the interval that starts at it, if the next detection's is usable; and every detection's instants start from its REFINED
        position s + f / 4 (the fraction that so far entered the period only)."""
        b = np.ascontiguousarray(b, dtype=F)
        n = len(b)
        L = int(spec["clk_lookahead"])
        tol, dmax, phases = 1 << int(spec["clk_tol_shift"]), 1 << int(spec["clk_dmax_log2"]), int(spec["clk_phases"])
        Ci = np.array(spec["clk_interp"], dtype=F).reshape(phases, 4)
        dets = [d for d in self.detections(b) if d[0] + self.W < n - L]
        # fraction of every sync position in quarter samples: vertex of the parabola through c[s - 1], c[s], c[s + 1]
        c = self._c
        fr = []
        for (s, _, _, _) in dets:
            cl, c0, cr = (c[s - 1] if s > 0 else F(0.0)), c[s], c[s + 1]
            num, den = F(cl - cr), F(F(cl - F(c0 + c0)) + cr)
            f = 0
            if den < 0:
                q = F(F(num * F(0.5)) / den)
                if q == q:
                    f = int(min(max(np.floor(F(F(q * F(4.0)) + F(0.5))), -2), 2))
            fr.append(f)
        ev = [(s + self.W + 1, 0, k) for k, (s, _, _, _) in enumerate(dets)] + [(min(max(int(q) - L, -L), n), 1, -1) for q in resync]
        ev.sort()
        bp = np.concatenate([np.zeros(2, dtype=F), b, np.zeros(4, dtype=F)])       # b[i] sits at bp[i + 2]
        # the backward clock of every detection, and whether its interval was usable (3.8b's test passed)
        back = {}
        prev = None
        for t, kind, k in ev:
            if kind == 0:
                s = dets[k][0]
                D, N, ok = self.sps, 1, False
                if prev is not None:
                    delta = s - prev[0]
                    Nn = (delta + 5) // 10
                    if Nn >= 1 and delta <= dmax and abs(delta - 10 * Nn) * tol <= 10 * Nn:
                        ok = True
                        d4 = 4 * delta + (fr[k] - prev[1])
                        if d4 != 40 * Nn:
                            D, N = d4, 4 * Nn
                back[k] = (D, N, ok)
                prev = (s, fr[k])
            else:
                prev = None
        order = [k for _, kind, k in ev if kind == 0]
        clock = {}
        for i, k in enumerate(order):
            D, N, ok = back[k]
            if reslice and not ok and i + 1 < len(order) and back[order[i + 1]][2]:
                D, N = back[order[i + 1]][0], back[order[i + 1]][1]
            off = 0
            if reslice:                                              # the fractional anchor: clocks are D / (4 N_sym), 10 / 1 is written 40 / 4
                if N == 1:
                    D, N = 4 * D, 4
                off = fr[k] * (N // 4)
            clock[k] = (D, N, off)
        dib, spos, sdib = [], [], []
        anchor, start = None, -L
        for t, kind, k in ev + [(n - L, 2, -1)]:
            t = min(t, n - L)
            if anchor is not None and t > start:
                s, hi, mid, lo, D, N, _, off = anchor
                # j with start <= s + (j D + off) div N < t, j >= 1
                j = max(1, ((start - s) * N - off + D - 1) // D - 1)
                while s + (j * D + off) // N < start:
                    j += 1
                js = []
                while s + (j * D + off) // N < t:
                    js.append(j); j += 1
                if js:
                    js = np.array(js, dtype=np.int64)
                    num = js * D + off
                    i = s + num // N
                    q = ((num % N) * phases) // N
                    w = Ci[q]
                    acc = (w[:, 0] * bp[i - 1 + 2]).astype(F)
                    acc = fma(w[:, 1], bp[i + 2], acc)
                    acc = fma(w[:, 2], bp[i + 1 + 2], acc)
                    acc = fma(w[:, 3], bp[i + 2 + 2], acc)
                    dib.append(np.where(acc >= hi, 1, np.where(acc >= mid, 0, np.where(acc >= lo, 2, 3))).astype(np.uint8))
            start = max(start, t)
            if kind == 0:
                s, hi, mid, lo = dets[k]
                D, N, off = clock[k]
                anchor = (s, hi, mid, lo, D, N, fr[k], off)
                spos.append(s)
                sdib.append(sum(len(x) for x in dib))
            elif kind == 1:
                anchor = None
        return (np.concatenate(dib) if dib else np.zeros(0, np.uint8), np.array(spos, dtype=np.int64), np.array(sdib, dtype=np.uint64))

    def receive(self, b, resync=()):
        """(dibits, sync_pos, sync_dibit) of SPEC 3.8 over the whole array; resync = sample indices q (lock dropped between
        samples q - 1 and q)."""
        b = np.ascontiguousarray(b, dtype=F)
        n = len(b)
        dets = self.detections(b)
        # events in the order they take effect: a detection governs instants > s + W, a drop removes instants >= q;
        # at the same instant the detection comes first (decided at q - 1) and is dropped with the rest
        ev = [(s + self.W + 1, 0, k) for k, (s, _, _, _) in enumerate(dets)] + [(min(max(int(q), 0), n), 1, -1) for q in resync]
        ev.sort()
        dib, spos, sdib = [], [], []
        anchor, start = None, 0
        bounds = ev + [(n, 2, -1)]
        for t, kind, k in bounds:
            if anchor is not None and t > start:
                s, hi, mid, lo = anchor
                m0 = max(1, -(-(start - s) // self.sps))              # first m with s + 10 m >= start
                idx = np.arange(s + self.sps * m0, min(t, n), self.sps)
                v = b[idx]
                d = np.where(v >= hi, 1, np.where(v >= mid, 0, np.where(v >= lo, 2, 3))).astype(np.uint8)
                dib.append(d)
            start = t
            if kind == 0:
                anchor = dets[k]
                spos.append(dets[k][0])
                sdib.append(sum(len(x) for x in dib))
            elif kind == 1:
                anchor = None
        return (np.concatenate(dib) if dib else np.zeros(0, np.uint8), np.array(spos, dtype=np.int64), np.array(sdib, dtype=np.uint64))


# ---------------------------------------------------------------------------------------------------------------------
# SPEC 3.1 - 3.5 as batch array operations (the oracle: streaming feed() objects with ring buffers, chunk by chunk).
# fma(a, b, c) is evaluated in a wider format and rounded once more to float32 (see fma() below): the product is exact, the sum
# rounds to the wide significand and then to 24 bits -- a double rounding that differs from a fused operation only when the
# wide result lands exactly on a float32 tie; demod() is expected to equal the oracle bit for bit.
# ---------------------------------------------------------------------------------------------------------------------
_WIDE = np.longdouble if np.finfo(np.longdouble).nmant >= 63 else np.float64     # x86: the 80-bit format, 64-bit significand


def fma(a, b, c):
    """fma of float32 operands: the product is exact in the wide format, the sum rounds to its significand and then to 24 bits.
    With float64 (53 bits) that second rounding moves a last bit about once in 2^29 operations -- tests/fuzz_oracle.py met it
    twice in 6 000 scenes (one and two baseband samples off by one ulp, the receiver's output unchanged) and the 80-bit format
    settled both in the oracle's favour; with 64 bits the odds are 2^-40 per operation."""
    return (np.asarray(a, dtype=_WIDE) * np.asarray(b, dtype=_WIDE) + np.asarray(c, dtype=_WIDE)).astype(F)


def _fir(x, taps, step, first):
    """acc = +0; for k: acc = fma(h[k], x[first + step m - k], acc) for every m with first + step m < len(x); x[n < 0] = 0"""
    n_out = (len(x) - first + step - 1) // step if len(x) > first else 0
    idx = first + step * np.arange(n_out)
    xp = np.concatenate([np.zeros(len(taps), dtype=F), x])          # index i of x sits at i + len(taps)
    acc = np.zeros(n_out, dtype=F)
    for k, h in enumerate(taps):
        acc = fma(F(h), xp[idx - k + len(taps)], acc)
    return acc


def atan2s(spec, y, x):
    c = [F(v) for v in spec["atan_coeffs"]]
    ax, ay = np.abs(x), np.abs(y)
    mx = np.where(ax > ay, ax, ay)
    mn = np.where(ax > ay, ay, ax)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (mn / mx).astype(F)
    s = (t * t).astype(F)
    p = np.full(len(t), c[-1], dtype=F)
    for i in range(len(c) - 2, -1, -1):
        p = fma(p, s, c[i])
    r = (p * t).astype(F)
    r = np.where(ay > ax, (F(spec["half_pi"]) - r).astype(F), r)
    r = np.where(x < 0, (F(spec["pi"]) - r).astype(F), r)
    r = np.where(y < 0, -r, r)
    return np.where(mx == 0, F(0.0), r).astype(F)


def padded_tables(spec, decim_taps=None, chan_taps=None):
    """SPEC 3.3: a configured table IS the table of the evaluation length -- 31 / 41, or 64 / 64 as soon as either is longer --
    with zero coefficients at the old end"""
    h1 = list(spec["decim_taps"] if decim_taps is None else decim_taps)
    h2 = list(spec["chan_taps"] if chan_taps is None else chan_taps)
    long_ = len(h1) > len(spec["decim_taps"]) or len(h2) > len(spec["chan_taps"])
    h1 += [0.0] * ((64 if long_ else len(spec["decim_taps"])) - len(h1))
    h2 += [0.0] * ((64 if long_ else len(spec["chan_taps"])) - len(h2))
    return h1, h2


def demod(spec, iq=None, u8=None, decim_taps=None, chan_taps=None, fm_gain=None, u8_scale=None, u8_offset=None, u8_lut=None,
          decim_phase=None, avg_taps=None):
    """48 kHz baseband of a whole capture (SPEC 3.1 - 3.5; the build's own numbers unless others are given: the same
    keywords as p25fe_config_t -- fm_gain is the resolved output scale)"""
    if u8 is not None:
        if u8_lut is not None:
            v = np.asarray(u8_lut, dtype=F)[np.asarray(u8, dtype=np.uint8)]
        else:
            b = np.asarray(u8, dtype=np.uint8).astype(F)
            v = fma(b, F(spec["u8_scale"] if u8_scale is None else u8_scale), F(spec["u8_offset"] if u8_offset is None else u8_offset))
        xr, xi = v[0::2], v[1::2]
    else:
        z = np.ascontiguousarray(iq, dtype=np.complex64)
        xr, xi = z.real.astype(F), z.imag.astype(F)
    h1, h2 = padded_tables(spec, decim_taps, chan_taps)
    dec = int(spec["decim"])
    ph = int(spec.get("decim_phase", dec - 1) if decim_phase is None else decim_phase)    # SPEC 3.2: output m from input 5 m + phase
    dr, di = _fir(xr, h1, dec, ph), _fir(xi, h1, dec, ph)
    yr, yi = _fir(dr, h2, 1, 0), _fir(di, h2, 1, 0)
    pr, pi_ = np.concatenate([[F(0)], yr[:-1]]), np.concatenate([[F(0)], yi[:-1]])        # y[m - 1], y[-1] = 0
    t = (yi * pi_).astype(F)
    re = fma(yr, pr, t)
    u = (yr * pi_).astype(F)
    im = fma(yi, pr, -u)
    fmv = (atan2s(spec, im, re) * F(spec["fm_gain"] if fm_gain is None else fm_gain)).astype(F)
    n = len(fmv)
    # SPEC 3.5: a table of equal taps is a moving average (summed newest first, one multiply); any other table a FIR
    h3 = np.asarray([spec["boxcar_scale"]] * int(spec["boxcar_len"]) if avg_taps is None else avg_taps, dtype=F)
    if len(set(h3.view(np.uint32).tolist())) > 1:
        return _fir(fmv, list(h3), 1, 0)
    L = len(h3)
    fp = np.concatenate([np.zeros(L, dtype=F), fmv])
    acc = fmv.copy()
    for j in range(1, L):
        acc = (acc + fp[L - j:L - j + n]).astype(F)
    return (acc * h3[0]).astype(F)


# ---------------------------------------------------------------------------------------------------------------------
# SPEC 3.9: the 65 536 code words of BCH(63,16,23) straight from the published generator polynomial (TIA-102.BAAA,
# 6331141367235453 octal) by polynomial division -- not from the generator-matrix rows tools/gen_spec.py writes.
# ---------------------------------------------------------------------------------------------------------------------
NID_GEN_OCTAL = "6331141367235453"


def nid_codewords():
    g = int(NID_GEN_OCTAL, 8)
    assert g.bit_length() == 48                                    # degree 47
    words = np.zeros(65536, dtype=np.uint64)
    for d in range(65536):
        r = d << 47
        for bit in range(62, 46, -1):                              # remainder of d x^47 modulo g
            if (r >> bit) & 1:
                r ^= g << (bit - 47)
        words[d] = (d << 47) | r
    return words


def nid_decode(codewords, word63):
    """(data, distance) of the nearest code word, smallest data word on ties"""
    x = codewords ^ np.uint64(word63)
    dist = np.zeros(len(x), dtype=np.int32)
    for sh in range(0, 64, 16):
        dist += _POP16[((x >> np.uint64(sh)) & np.uint64(0xffff)).astype(np.int64)]
    k = int(np.argmin(dist))                                        # argmin returns the first (smallest data word)
    return k, int(dist[k])


_POP16 = np.array([bin(i).count("1") for i in range(65536)], dtype=np.int32)
