#!/usr/bin/env python3
"""Ahead-of-time kernel specialisation (the `make SPEC=file.json` step): compile the front-end kernels for a set of numbers
and store the code object where a deployment's p25fe_create will find it ($P25FE_SPEC_DIR).  Needs no GPU.

    tools/specialize.py numbers.json [-o DIR]
    make -C p25rx_amd/csrc spec SPEC=numbers.json [SPEC_DIR=DIR]

numbers.json: any of the p25fe_config_t fields by name -- decim_taps, chan_taps (lists), fm_deviation_hz, fm_sample_rate_hz,
fm_gain, u8_scale, u8_offset, u8_lut (256 values), avg_taps (list), decim_phase -- or the consts.json of tools/pin/dump_consts.rs (recognised by its
"chan_impulse" key: the reference's own tables, LUT and discriminator scale).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("numbers")
    ap.add_argument("-o", "--out", default=None, help="directory (default: the library's cache directory)")
    a = ap.parse_args()
    with open(a.numbers) as f:
        raw = json.load(f)
    if "chan_impulse" in raw:
        sys.path.insert(0, os.path.join(ROOT, "tools", "pin"))
        import load_pin
        kw = load_pin.config_kwargs(load_pin.load_consts(a.numbers))
    else:
        allowed = ("decim_taps", "chan_taps", "fm_deviation_hz", "fm_sample_rate_hz", "fm_gain", "u8_scale", "u8_offset", "u8_lut", "avg_taps",
                   "decim_phase")
        unknown = set(raw) - set(allowed)
        if unknown:
            sys.exit("unknown keys: %s (allowed: %s)" % (sorted(unknown), ", ".join(allowed)))
        kw = raw
    from p25rx_amd import _lib
    path = _lib.specialize(_lib.make_config(**kw), a.out)
    print(path if path else "these are the library's own numbers: its built-in kernels carry them, nothing to store")


if __name__ == "__main__":
    main()
