#!/usr/bin/env python3
"""CPU-only randomised cross-check of the oracle against the independent batch model (tests/spec_model.py): baseband bit
for bit (cf32 and u8), receiver dibit for dibit under both symbol clocks, random chunkings and lock drops.

    python tests/fuzz_oracle.py [--cases N] [--seed S]            (test infrastructure; 1 500 scenes take about a minute)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=90000)
    a = ap.parse_args()
    from oracle import oracle as O
    import spec_model
    from p25rx_amd import c4fm
    spec = O.load_spec()
    t0, tot, bad = time.time(), 0, 0
    for seed in range(a.seed, a.seed + a.cases):
        rng = np.random.default_rng(seed)
        mode = int(rng.integers(0, 2))
        snr = float(rng.choice([30.0, 12.0, 6.0, 3.0, 0.0]))
        frame = int(rng.choice([24, 30, 48, 100, 864, 3000]))
        ppm = float(rng.choice([0.0, 40.0, -100.0, 250.0])) if mode else 0.0
        iq, _, _ = c4fm.synth(float(rng.choice([0.1, 0.3, 0.7])), seed=seed, snr_db=snr, frame_dibits=frame,
                              freq_offset_hz=float(rng.choice([0.0, 300.0, -800.0])), timing_offset=int(rng.integers(0, 50)),
                              amplitude=float(rng.choice([0.5, 0.05])), clock_ppm=ppm)
        dt = ct = None
        if rng.integers(0, 5) == 0:                                  # caller-supplied tables of random length (SPEC 3.3's padding rule)
            big = bool(rng.integers(0, 2))
            dt = [float(np.float32(v)) for v in rng.normal(0, 0.2, int(rng.integers(32, 65) if big else rng.integers(1, 32)))]
            ct = [float(np.float32(v)) for v in rng.normal(0, 0.2, int(rng.integers(1, 65) if big else rng.integers(1, 42)))]
        d = O.Demod(O.make_config(decim_taps=dt, chan_taps=ct))
        if rng.integers(0, 2):
            raw = c4fm.to_u8(iq)
            bb, ref_bb = d.feed_u8(raw), spec_model.demod(spec, u8=raw, decim_taps=dt, chan_taps=ct)
        else:
            bb, ref_bb = d.feed_cf32(iq), spec_model.demod(spec, iq=iq, decim_taps=dt, chan_taps=ct)
        if not np.array_equal(bb.view(np.uint32), ref_bb.view(np.uint32)):
            bad += 1
            print("baseband differs: seed", seed)
        drops = sorted(set(int(x) for x in rng.integers(0, len(bb) + 3, size=int(rng.choice([0, 0, 2, 15])))))
        r = O.Recv(O.make_config(symbol_clock=mode))
        outs, o = [], 0
        for q in sorted(set(drops + [int(x) for x in rng.integers(0, len(bb), size=6)] + [len(bb)])):
            q = min(q, len(bb))
            outs.append(r.feed(bb[o:q]))
            if q in drops:
                r.resync()
            o = q
        got = [np.concatenate([x[k] for x in outs]) for k in range(3)]
        m = spec_model.Model(spec)
        ref = m.receive_tracking(bb, spec, drops) if mode else m.receive(bb, drops)
        if not all(len(got[k]) == len(ref[k]) and np.array_equal(got[k], ref[k].astype(got[k].dtype)) for k in range(3)):
            bad += 1
            print("receiver differs: seed", seed, "mode", mode, "margin", m.margins())
        tot += len(ref[0])
    print("fuzz_oracle: %d scenes, %d dibits, %d differences, %.0f s" % (a.cases, tot, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
