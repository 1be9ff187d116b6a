#!/usr/bin/env python3
"""Turn the reference's dumps (tools/pin/dump_consts.rs, dump_golden.rs) into the numbers this repository runs with.

load_consts(path) -> dict:
    decim_taps, chan_taps   float32 arrays, tap 0 multiplies the newest sample (docs/SPEC.md 3.2 / 3.3)
    decim_phase             index (0..4) of the input sample that produces the first output of a fresh decimator
                            (docs/SPEC.md 3.2 assumes 4: outputs at the 5th, 10th, ... sample)
    u8_lut                  256 float32: the value of a byte (the same table for I and Q, src/demod.rs:83)
    boxcar_len, boxcar_scale
    fm_gain                 float32 output scale of FmDemod::new(5000, 48000) that best explains the probes
    fm_probe_max_err, fm_pairs_max_err   |reference - oracle discriminator with that scale| over the dumped probes
    standin                 True when the file was written by tools/pin/make_standin.py (it pins NOTHING)
config_kwargs(consts) -> the keyword arguments shared by p25rx_amd._lib.make_config / FrontEnd and oracle.make_config.
"""
import json
import re

import numpy as np


def _f32(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def load_consts(path):
    with open(path) as f:
        raw = json.load(f)
    out = {"standin": bool(raw.get("standin", False))}
    # ---- rtlsdr_iq::IQ: 65536 x (re, im); re must depend on the low byte only, im on the high byte only, same table
    lut = np.asarray(raw["iq_lut"], dtype=np.uint32).reshape(65536, 2)
    re = lut[:, 0].reshape(256, 256)            # [high byte][low byte]
    im = lut[:, 1].reshape(256, 256)
    if not (np.all(re == re[0:1, :]) and np.all(im == im[:, 0:1])):
        raise ValueError("IQ[s] is not a per-byte table: the first byte is not I / the second not Q (docs/SPEC.md 3.1)")
    if not np.array_equal(re[0, :], im[:, 0]):
        raise ValueError("IQ[s] uses different tables for I and Q")
    out["u8_lut"] = re[0, :].copy().view(np.float32)
    # ---- channel filter: the impulse response IS the table (y[n] = sum h[k] x[n - k], one fma per tap: h[k] * 1 exactly)
    ch = _f32(raw["chan_impulse"])
    nz = np.nonzero(ch)[0]
    out["chan_taps"] = ch[:nz[-1] + 1].copy()
    # ---- decimator: impulse at input p -> output m carries h[5 m + phase - p]
    imp = [_f32(r) for r in raw["decim_impulse"]]
    first = [p for p in range(min(5, len(imp))) if len(imp[p]) and imp[p][0] != 0.0]
    if not first:
        raise ValueError("no impulse position produces a first output: cannot find the decimator's phase")
    phase = max(first)
    taps = {}
    for p, row in enumerate(imp):
        for m, v in enumerate(row):
            k = 5 * m + phase - p
            if k >= 0 and v != 0.0:
                if k in taps and taps[k] != v:
                    raise ValueError("decimator impulse responses disagree at tap %d" % k)
                taps[k] = v
    n1 = max(taps) + 1
    out["decim_taps"] = np.array([taps.get(k, np.float32(0)) for k in range(n1)], dtype=np.float32)
    out["decim_phase"] = int(phase)
    # ---- moving average
    av = _f32(raw["avg_impulse"])
    nz = np.nonzero(av)[0]
    out["boxcar_len"] = int(nz[-1] + 1)
    out["boxcar_scale"] = float(av[0])
    out["avg_taps"] = av[:nz[-1] + 1].copy()                       # the impulse response IS the table (p25fe_config_t.avg_taps, ABI 5)
    # ---- discriminator: output = angle * gain; choose the float32 gain that explains the probes best
    pr = np.asarray(raw["fm_probe"], dtype=np.int64)
    k = pr[:, 0]
    y = pr[:, 1].astype(np.uint32).view(np.float32).astype(np.float64)
    ang = np.where(k <= 180, k, k - 360) * (2.0 * np.pi / 360.0)
    sel = (k > 0) & (k < 90)
    g_ls = float(np.sum(y[sel] * ang[sel]) / np.sum(ang[sel] ** 2))
    cands = {np.float32(48000.0 / (2.0 * np.pi * 5000.0)),
             np.float32(np.float32(48000.0) / (np.float32(2.0) * np.float32(np.pi) * np.float32(5000.0))), np.float32(g_ls)}
    best = min(cands, key=lambda g: float(np.max(np.abs(y[sel] - ang[sel] * float(g)))))
    out["fm_gain"] = float(best)
    out["fm_gain_least_squares"] = g_ls
    out["fm_probe_max_err"] = float(np.max(np.abs(y[(k != 180)] - ang[(k != 180)] * float(best))))
    out["fm_pairs"] = np.asarray(raw.get("fm_pairs", []), dtype=np.uint32).reshape(-1, 5)
    return out


def config_kwargs(consts):
    """keyword arguments for _lib.make_config / FrontEnd / oracle.make_config"""
    return dict(decim_taps=[float(x) for x in consts["decim_taps"]], chan_taps=[float(x) for x in consts["chan_taps"]],
                u8_lut=np.asarray(consts["u8_lut"], dtype=np.float32), fm_gain=float(consts["fm_gain"]),
                decim_phase=int(consts["decim_phase"]), avg_taps=[float(x) for x in consts["avg_taps"]])


DUID = {"VoiceHeader": 0x0, "VoiceSimpleTerminator": 0x3, "VoiceLCFrameGroup": 0x5, "TrunkingSignaling": 0x7,
        "VoiceCCFrameGroup": 0xA, "DataPacket": 0xC, "VoiceLCTerminator": 0xF}
NAC = {"Default": 0x293, "ReceiveAny": 0xF7E, "RepeatAny": 0xF7F}


def load_nid_log(path):
    """[(sample index, nac or None, duid or None)] from dump_golden's lines ({:?} of p25's NetworkId, or plain nac / duid numbers)"""
    ev = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            d = json.loads(line)
            nac, duid = d.get("nac"), d.get("duid")
            txt = d.get("nid", "")
            m = re.search(r"data_unit:\s*(\w+)", txt)
            if m and duid is None:
                duid = DUID.get(m.group(1))
            m = re.search(r"access_code:\s*(\w+)(?:\((0x[0-9a-fA-F]+|\d+)\))?", txt)
            if m and nac is None:
                nac = int(m.group(2), 0) if m.group(2) else NAC.get(m.group(1))
            ev.append((int(d["sample"]), nac, duid))
    return ev


if __name__ == "__main__":
    import sys
    c = load_consts(sys.argv[1])
    print("stand-in (pins nothing)" if c["standin"] else "reference dump")
    print("decimator: %d taps, first output from input %d (SPEC assumes 4)" % (len(c["decim_taps"]), c["decim_phase"]))
    print("channel filter: %d taps; boxcar %d x %.9g" % (len(c["chan_taps"]), c["boxcar_len"], c["boxcar_scale"]))
    print("fm gain %.9g (least squares %.9g), probe error %.3g" % (c["fm_gain"], c["fm_gain_least_squares"], c["fm_probe_max_err"]))
    lut = c["u8_lut"]
    print("u8 table: lut[0] %.9g lut[127] %.9g lut[128] %.9g lut[255] %.9g" % (lut[0], lut[127], lut[128], lut[255]))
