#!/usr/bin/env python3
"""Experiment (numpy, CPU; not part of the product or the suite): the prototype sweep behind SPEC 3.8b's period rule -- the
PERIOD taken from sync positions refined to 1 / `sub` sample (vertex of the parabola through the correlation peak and its two
neighbours), the anchor itself staying on the whole sample.  It showed that this removes the small-ppm penalty of the
whole-sample estimate and that quarter samples are enough; the rule is now built (oracle, tests/spec_model.py, kernels), so
the middle column below (SPEC 3.8b as built) equals the `sub = 4` one.  Prints symbol errors against the modulator."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O                      # noqa: E402
import spec_model                                   # noqa: E402
from p25rx_amd import c4fm                          # noqa: E402

F = np.float32


def receive_refined(m, b, spec, sub=8):
    b = np.ascontiguousarray(b, dtype=F)
    n = len(b)
    L = int(spec["clk_lookahead"])
    tol, dmax, phases = 1 << int(spec["clk_tol_shift"]), 1 << int(spec["clk_dmax_log2"]), int(spec["clk_phases"])
    Ci = np.array(spec["clk_interp"], dtype=F).reshape(phases, 4)
    dets = [d for d in m.detections(b) if d[0] + m.W < n - L]
    c = m._c
    fr = []
    for (s, _, _, _) in dets:
        cl, c0, cr = c[s - 1] if s > 0 else F(0), c[s], c[s + 1]
        den = F(F(cl - F(2) * c0) + cr)
        f = F(F(0.5) * F(cl - cr) / den) if den != 0 else F(0)
        fr.append(int(np.clip(np.floor(float(f) * sub + 0.5), -sub // 2, sub // 2)))
    bp = np.concatenate([np.zeros(2, dtype=F), b, np.zeros(4, dtype=F)])
    dib, anchor, start = [], None, -L
    ev = [(s + m.W + 1, k) for k, (s, _, _, _) in enumerate(dets)] + [(n - L, -1)]
    for t, k in ev:
        t = min(t, n - L)
        if anchor is not None and t > start:
            s, hi, mid, lo, D, N, _ = anchor
            j = max(1, ((start - s) * N + D - 1) // D)
            while s + (j * D) // N < start:
                j += 1
            js = []
            while s + (j * D) // N < t:
                js.append(j); j += 1
            if js:
                js = np.array(js, dtype=np.int64)
                num = js * D
                i = s + num // N
                q = ((num % N) * phases) // N
                w = Ci[q]
                acc = (w[:, 0] * bp[i - 1 + 2]).astype(F)
                acc = spec_model.fma(w[:, 1], bp[i + 2], acc)
                acc = spec_model.fma(w[:, 2], bp[i + 1 + 2], acc)
                acc = spec_model.fma(w[:, 3], bp[i + 2 + 2], acc)
                dib.append(np.where(acc >= hi, 1, np.where(acc >= mid, 0, np.where(acc >= lo, 2, 3))).astype(np.uint8))
        start = max(start, t)
        if k >= 0:
            s, hi, mid, lo = dets[k]
            D, N = m.sps, 1
            if anchor is not None:
                delta = s - anchor[0]
                Nn = (delta + 5) // 10
                if Nn >= 1 and delta <= dmax and abs(delta - 10 * Nn) * tol <= 10 * Nn:
                    D, N = sub * delta + (fr[k] - anchor[6]), sub * Nn
            anchor = (s, hi, mid, lo, D, N, fr[k])
    return np.concatenate(dib) if dib else np.zeros(0, np.uint8)


def main():
    spec = O.load_spec()
    print("ppm   frame  fixed  spec-3.8b  refined-1/8  (symbol errors of n)")
    for frame, secs in ((864, 6.0), (3000, 8.0)):
        for ppm in (0.0, 20.0, -20.0, 60.0, 100.0, 150.0, 250.0):
            iq, truth, _ = c4fm.synth(secs, seed=5, snr_db=30.0, clock_ppm=ppm, frame_dibits=frame)
            bb = O.Demod().feed_cf32(iq)
            m = spec_model.Model(spec)
            outs = [m.receive(bb)[0], m.receive_tracking(bb, spec)[0], receive_refined(m, bb, spec)]
            errs = []
            for d in outs:
                k = min(len(d), len(truth) - 24)
                errs.append(int(np.count_nonzero(d[:k] != truth[24:24 + k])))
            print("%5.0f %6d %6d %9d %11d   n = %d" % (ppm, frame, errs[0], errs[1], errs[2], k))


if __name__ == "__main__":
    main()
