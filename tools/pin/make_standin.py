#!/usr/bin/env python3
"""STAND-IN for the reference's dumps, written from THIS repository's own oracle.  It pins nothing -- it exists so that the
pin harness (tools/pin/load_pin.py, tests/test_pin.py) is exercised end to end without a Rust toolchain: the files have the
exact format tools/pin/dump_consts.rs / dump_golden.rs write and carry "standin": true.

usage: make_standin.py OUTDIR [--variant]      (--variant: other tables / constants than the build's, to prove the loader
                                                recovers what is in the file and not what the build already has)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def bits(a):
    return [int(x) for x in np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).reshape(-1)]


def pin_capture():
    """the capture tools/pin/README.md tells the maintainer to feed the reference: 2 s, seed 7, every frame a terminator NID"""
    from p25rx_amd import c4fm
    iq, dibits, _ = c4fm.synth(2.0, seed=7, snr_db=20.0, nid=lambda f: (0x293, 0x3))
    return c4fm.to_u8(iq), dibits


def main():
    outdir = sys.argv[1]
    variant = "--variant" in sys.argv[2:]
    os.makedirs(outdir, exist_ok=True)
    from oracle import oracle as O
    spec = O.load_spec()
    kw = {}
    if variant:
        from scipy import signal as sps
        kw = dict(decim_taps=sps.firwin(37, 11000.0, window=("kaiser", 6.0), fs=240000.0).astype(np.float32).tolist(),
                  chan_taps=sps.firwin(48, 6500.0, window=("kaiser", 4.5), fs=48000.0).astype(np.float32).tolist(),
                  u8_lut=((np.arange(256, dtype=np.float64) - 127.5) / 127.5).astype(np.float32),
                  fm_gain=float(np.float32(np.float32(48000.0) / (np.float32(2.0) * np.float32(np.pi) * np.float32(5000.0)))),
                  # ABI 5: another decimator phase and a post-discriminator filter that is no moving average
                  decim_phase=2, avg_taps=(np.hanning(23)[1:-1] / np.hanning(23)[1:-1].sum()).astype(np.float32).tolist())
    cfg = O.make_config(spec, **kw)
    lut = np.array([cfg.u8_lut[b] if cfg.u8_lut_valid else np.float32(np.float64(np.float32(b)) * np.float64(np.float32(cfg.u8_scale)) + np.float64(cfg.u8_offset))
                    for b in range(256)], dtype=np.float32)
    iq_lut = [[int(lut[s & 255].view(np.uint32)), int(lut[s >> 8].view(np.uint32))] for s in range(65536)]
    # the behavioural probes, answered by the oracle's objects
    decim_imp = []
    for p in range(10):
        x = np.zeros(640, dtype=np.complex64)
        x[p] = 1.0
        ch, fm, bb = O.Demod(O.make_config(spec, decim_taps=kw.get("decim_taps"), chan_taps=[1.0], decim_phase=kw.get("decim_phase"))).feed_cf32_stages(x)
        decim_imp.append(bits(ch.real))                          # channel filter = identity: the decimator's outputs
    x = np.zeros(5 * 256, dtype=np.complex64)
    x[4] = 1.0                                                   # one decimator output of exactly 1 needs a one-tap decimator
    ch, _, _ = O.Demod(O.make_config(spec, decim_taps=[1.0], chan_taps=kw.get("chan_taps"))).feed_cf32_stages(x)
    chan_imp = bits(ch.real[:256])
    avg = np.zeros(96, dtype=np.float32)                         # the post-discriminator filter's impulse response
    avg[:cfg.n_avg] = np.array([cfg.avg_taps[k] for k in range(cfg.n_avg)], dtype=np.float32)
    probes, pairs = [], []
    for k in range(360):
        a = 2.0 * np.pi * k / 360.0
        d = O.Demod(O.make_config(spec, decim_taps=[1.0], chan_taps=[1.0], fm_gain=kw.get("fm_gain")))
        z = np.zeros(10, dtype=np.complex64)
        z[4] = 1.0
        z[9] = np.complex64(complex(np.float32(np.cos(a)), np.float32(np.sin(a))))
        _, fm, _ = d.feed_cf32_stages(z)
        probes.append([k, bits(fm[1:2])[0]])
    st = np.uint32(0x2545F491)

    def nxt():
        nonlocal st
        with np.errstate(over="ignore"):
            st ^= np.uint32(st << np.uint32(13))
            st ^= np.uint32(st >> np.uint32(17))
            st ^= np.uint32(st << np.uint32(5))
        return np.float32(float(st) / 4294967296.0 - 0.5)
    for _ in range(512):
        pr, pi, cr, ci = nxt(), nxt(), nxt(), nxt()
        d = O.Demod(O.make_config(spec, decim_taps=[1.0], chan_taps=[1.0], fm_gain=kw.get("fm_gain")))
        z = np.zeros(10, dtype=np.complex64)
        z[4] = np.complex64(complex(pr, pi))
        z[9] = np.complex64(complex(cr, ci))
        _, fm, _ = d.feed_cf32_stages(z)
        pairs.append(bits([pr, pi, cr, ci]) + bits(fm[1:2]))
    with open(os.path.join(outdir, "consts.json"), "w") as f:
        json.dump({"standin": True, "note": "written by tools/pin/make_standin.py from this repository's oracle: pins nothing",
                   "iq_lut": iq_lut, "decim_impulse": decim_imp, "chan_impulse": chan_imp, "avg_impulse": bits(avg),
                   "fm_probe": probes, "fm_pairs": pairs}, f)
    # the golden pair: capture, baseband in the reference's 32 768-byte buffers, NID log
    u8, _ = pin_capture()
    u8.tofile(os.path.join(outdir, "pin_seed7.u8"))
    d = O.Demod(cfg)
    bb = np.concatenate([d.feed_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
    bb.tofile(os.path.join(outdir, "baseband.f32le"))
    dib, spos, sdib = O.Recv(cfg).feed(bb)
    nid = O.nid_decode(dib, sdib, spos)
    names = {0x0: "VoiceHeader", 0x3: "VoiceSimpleTerminator", 0x5: "VoiceLCFrameGroup", 0x7: "TrunkingSignaling",
             0xA: "VoiceCCFrameGroup", 0xC: "DataPacket", 0xF: "VoiceLCTerminator"}
    with open(os.path.join(outdir, "nid.jsonl"), "w") as f:
        for r in nid:
            if int(r["valid"]) == 1 and int(r["duid"]) in names:
                ac = "Default" if int(r["nac"]) == 0x293 else "Other(%d)" % int(r["nac"])
                # (the reference reports a NID when its last dibit has arrived: 33 dibits after the sync word's last symbol)
                f.write(json.dumps({"sample": int(r["sync_pos"]) + 330, "nid": "NetworkId { access_code: %s, data_unit: %s }"
                                    % (ac, names[int(r["duid"])])}) + "\n")


if __name__ == "__main__":
    main()
