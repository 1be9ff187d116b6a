"""The pin harness: parity against the REAL kchmck/p25rx, ready for the day someone with `cargo` produces the dumps.

tools/pin/dump_consts.rs and dump_golden.rs (sources only: no Rust toolchain in this image) write consts.json, baseband.f32le
and nid.jsonl; `tools/pin/README.md` is the one-command procedure.  With those files under tests/golden/pin/ the tests marked
`pin` load the reference's tables / LUT / discriminator scale into the oracle AND (on the GPU box) into the library through
p25fe_config_t, run both on pin_seed7.u8 and compare with what the reference produced.  Until then they are skipped and
parity stays UNPINNED.  What always runs is the same harness on a STAND-IN written from this repository's own oracle
(tools/pin/make_standin.py) -- it pins nothing; it proves that the loader recovers the numbers that are in a dump (not the
ones the build already has) and that the comparison code works.
"""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_DIR = os.path.join(ROOT, "tests", "golden", "pin")
# docs/SPEC.md 3.4: the oracle's atan2 is a polynomial (two compilers agree bit for bit), the reference's is libm's: <= 4e-7 rad
# per discriminator output, times the scale 1.53, averaged over 10 samples -- plus summation-order differences of the FIRs
BASEBAND_TOL = 2e-6


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "pin", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def compare_with_dump(d, gpu):
    """Run the oracle (and, when gpu, the HIP path through the C ABI) with the dump's numbers on the dump's capture; returns a
    report dict.  Raises on a structural mismatch; tolerances are applied by the callers."""
    from oracle import oracle as O
    lp = _load("load_pin")
    c = lp.load_consts(os.path.join(d, "consts.json"))
    kw = lp.config_kwargs(c)
    rep = {"standin": c["standin"], "decim_phase": c["decim_phase"], "n_decim": len(c["decim_taps"]), "n_chan": len(c["chan_taps"]),
           "boxcar": (c["boxcar_len"], c["boxcar_scale"]), "fm_gain": c["fm_gain"], "fm_probe_max_err": c["fm_probe_max_err"]}
    # (ABI 5: whatever phase the reference's decimator has and whatever its MovingAverage's impulse response is travel as DATA --
    # p25fe_config_t.decim_phase / avg_taps -- instead of being asserted equal to docs/SPEC.md's 4 and ten taps of 0.1)
    ocfg = O.make_config(None, **kw)
    # the discriminator, value by value, on the dumped pairs
    if len(c["fm_pairs"]):
        p = c["fm_pairs"]
        prev = p[:, 0:2].copy().view(np.float32)
        cur = p[:, 2:4].copy().view(np.float32)
        want = p[:, 4].copy().view(np.float32)
        got = np.empty(len(p), dtype=np.float32)
        one = O.make_config(None, decim_taps=[1.0], chan_taps=[1.0], fm_gain=kw["fm_gain"])
        for i in range(len(p)):
            z = np.zeros(10, dtype=np.complex64)
            z[4] = complex(prev[i, 0], prev[i, 1])
            z[9] = complex(cur[i, 0], cur[i, 1])
            got[i] = O.Demod(one).feed_cf32_stages(z)[1][1]
        rep["fm_pairs_max_err"] = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
    u8 = np.fromfile(os.path.join(d, "pin_seed7.u8"), dtype=np.uint8)
    ref_bb = np.fromfile(os.path.join(d, "baseband.f32le"), dtype=np.float32)
    dm = O.Demod(ocfg)
    bb = np.concatenate([dm.feed_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
    assert len(bb) == len(ref_bb), (len(bb), len(ref_bb))
    rep["baseband_max_err"] = float(np.max(np.abs(bb.astype(np.float64) - ref_bb.astype(np.float64))))
    rep["baseband_bit_exact"] = bool(np.array_equal(bb.view(np.uint32), ref_bb.view(np.uint32)))
    dib, spos, sdib = O.Recv(ocfg).feed(bb)
    nid = O.nid_decode(dib, sdib, spos)
    mine = [(int(r["nac"]), int(r["duid"])) for r in nid if int(r["valid"]) == 1]
    theirs = lp.load_nid_log(os.path.join(d, "nid.jsonl"))
    rep["nid_mine"], rep["nid_theirs"] = mine, [(n, u) for _, n, u in theirs]
    if gpu:
        from p25rx_amd.frontend import FrontEnd
        fe = FrontEnd(specialize=1, **kw)                          # the reference's numbers as immediates: kernels specialised for them
        got = np.concatenate([fe.demod_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
        rep["gpu_equals_oracle"] = bool(np.array_equal(got.view(np.uint32), bb.view(np.uint32)))
        rep["gpu_baseband_max_err"] = float(np.max(np.abs(got.astype(np.float64) - ref_bb.astype(np.float64))))
        fe.reset()
        gd = np.concatenate([fe.run_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
        rep["gpu_dibits_equal_oracle"] = bool(np.array_equal(gd, dib))
        rep["gpu_variant"] = fe.kernel_variant
    return rep


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pin_standin"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pin", "make_standin.py"), d, "--variant"])
    return d


def test_pin_harness_recovers_what_is_in_a_dump(standin):
    """STAND-IN (pins nothing): the loader must recover the variant tables / LUT / scale that the stand-in was written with --
    none of them the build's own -- from impulse responses and probes alone, and the comparison must come out exact."""
    from scipy import signal as sps
    lp = _load("load_pin")
    c = lp.load_consts(os.path.join(standin, "consts.json"))
    assert c["standin"] is True
    want_d = sps.firwin(37, 11000.0, window=("kaiser", 6.0), fs=240000.0).astype(np.float32)
    want_c = sps.firwin(48, 6500.0, window=("kaiser", 4.5), fs=48000.0).astype(np.float32)
    assert np.array_equal(c["decim_taps"].view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(c["chan_taps"].view(np.uint32), want_c.view(np.uint32))
    assert np.array_equal(c["u8_lut"].view(np.uint32), ((np.arange(256, dtype=np.float64) - 127.5) / 127.5).astype(np.float32).view(np.uint32))
    g = np.float32(np.float32(48000.0) / (np.float32(2.0) * np.float32(np.pi) * np.float32(5000.0)))
    want_a = (np.hanning(23)[1:-1] / np.hanning(23)[1:-1].sum()).astype(np.float32)
    assert np.float32(c["fm_gain"]) == g and c["decim_phase"] == 2 and c["boxcar_len"] == 21
    assert np.array_equal(np.asarray(c["avg_taps"], dtype=np.float32).view(np.uint32), want_a.view(np.uint32))
    rep = compare_with_dump(standin, gpu=False)
    assert rep["baseband_bit_exact"] and rep["fm_pairs_max_err"] == 0.0
    assert rep["nid_mine"] == rep["nid_theirs"] and len(rep["nid_mine"]) >= 10 and all(x == (0x293, 0x3) for x in rep["nid_mine"])


def test_ahead_of_time_specialisation_from_a_dump(standin, tmp_path):
    """`make spec SPEC=consts.json SPEC_DIR=dir` (tools/specialize.py): the dump's numbers -> the code object a deployment's
    p25fe_create finds through $P25FE_SPEC_DIR.  No GPU needed; the file name is the one p25fe_specialize gives for the same config."""
    from p25rx_amd import _lib
    d = str(tmp_path / "aot")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "specialize.py"), os.path.join(standin, "consts.json"), "-o", d],
                                  text=True).strip()
    assert os.path.dirname(out) == d and os.path.exists(out)
    lp = _load("load_pin")
    kw = lp.config_kwargs(lp.load_consts(os.path.join(standin, "consts.json")))
    assert _lib.specialize(_lib.make_config(**kw), d) == out
    js = tmp_path / "numbers.json"
    js.write_text('{"fm_deviation_hz": 5000, "u8_offset": -1.0}')     # the build's own numbers: nothing to compile
    msg = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "specialize.py"), str(js), "-o", d], text=True)
    assert "own numbers" in msg


@pytest.mark.gpu
def test_pin_harness_gpu_leg_on_the_stand_in(standin):
    """STAND-IN (pins nothing): the dump's numbers reach the GPU through p25fe_config_t (ABI 4: tables, the u8 table, the
    discriminator's scale) as specialised kernels, and the HIP path equals the oracle configured from the same dump."""
    rep = compare_with_dump(standin, gpu=True)
    assert rep["gpu_equals_oracle"] and rep["gpu_dibits_equal_oracle"] and rep["gpu_variant"] == 1


def _have_dump():
    return all(os.path.exists(os.path.join(PIN_DIR, f)) for f in ("consts.json", "pin_seed7.u8", "baseband.f32le", "nid.jsonl"))


@pytest.mark.pin
@pytest.mark.skipif(not _have_dump(), reason="tests/golden/pin/ holds no dump of the reference (tools/pin/README.md): parity UNPINNED")
def test_pin_oracle_against_the_reference_dump():
    rep = compare_with_dump(PIN_DIR, gpu=False)
    assert not rep["standin"], "tests/golden/pin/ holds a stand-in, not a dump of the reference"
    assert rep["baseband_max_err"] <= BASEBAND_TOL, rep
    assert rep["nid_mine"] == rep["nid_theirs"], rep


@pytest.mark.pin
@pytest.mark.gpu
@pytest.mark.skipif(not _have_dump(), reason="tests/golden/pin/ holds no dump of the reference (tools/pin/README.md): parity UNPINNED")
def test_pin_gpu_against_the_reference_dump():
    rep = compare_with_dump(PIN_DIR, gpu=True)
    assert not rep["standin"]
    assert rep["gpu_equals_oracle"] and rep["gpu_dibits_equal_oracle"]
    assert rep["gpu_baseband_max_err"] <= BASEBAND_TOL, rep
