"""GPU parity of the reference's REAL configuration surface: caller-supplied tap tables (p25_filts' DecimFir / BandpassFir,
src/demod.rs:27-29), the arguments of FmDemod::new (src/demod.rs:54) and the rtlsdr_iq table (src/demod.rs:83) -- through
all three kernel variants of the library: the built-in immediate-coefficient kernels (the build's own numbers), kernels
specialised for the caller's numbers (hipRTC at p25fe_create + on-disk cache) and the generic LDS-tap fallback.
Every comparison is bit for bit against the oracle configured with the same numbers.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def lib():
    from p25rx_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def FE():
    from p25rx_amd.frontend import FrontEnd
    return FrontEnd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rand_taps(rng, n):
    return (rng.standard_normal(n) * np.hanning(n + 2)[1:-1] / 6).astype(np.float32).tolist()


def all_paths(O, FE, lib, iq, ocfg_kw, fe_kw, want_variant, u8=None):
    """One handle configuration through every front-end kernel shape: linear K1 (demod_*: the RecvEvent::Baseband hand-off),
    planar K1 (run_dev), the one-launch chunk kernel (run_* on 16 384-sample chunks), cf32 and u8."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    spec = O.load_spec()
    ocfg = O.make_config(spec, **ocfg_kw)
    u8 = c4fm.to_u8(iq) if u8 is None else u8
    ref = O.Demod(ocfg).feed_cf32(iq)
    ref8 = O.Demod(ocfg).feed_u8(u8)
    fe = FE(**fe_kw)
    assert fe.kernel_variant == want_variant, (fe.kernel_variant, want_variant, lib.specialize_log()[-500:])
    cuts = [0, 5, 16384, 16385, 60001, len(iq)]
    got = np.concatenate([fe.demod_cf32(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(bits(got), bits(ref))
    fe.reset()
    got8 = np.concatenate([fe.demod_u8(u8[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(bits(got8), bits(ref8))
    # planar K1 + receiver on a resident capture, both formats
    dib_ref = O.Recv(ocfg).feed(ref)[0]
    dib_ref8 = O.Recv(ocfg).feed(ref8)[0]
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    dib, res = fe.run_dev(t)
    n = int(parse_results(res)[0]["n_dibits"])
    assert n == len(dib_ref) and np.array_equal(dib[0, :n].cpu().numpy(), dib_ref)
    t8 = torch.from_numpy(u8.reshape(-1, 2)).cuda()
    dib, res = fe.run_dev(t8)
    n = int(parse_results(res)[0]["n_dibits"])
    assert n == len(dib_ref8) and np.array_equal(dib[0, :n].cpu().numpy(), dib_ref8)
    # the reference's own chunking through the streaming call: one launch per chunk (k_chunk)
    fe.reset()
    got = np.concatenate([fe.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
    assert np.array_equal(got, dib_ref)
    fe.reset()
    got = np.concatenate([fe.run_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
    assert np.array_equal(got, dib_ref8)
    return fe


@pytest.mark.parametrize("nt", [(31, 41), (19, 33), (64, 64), (40, 41)])
def test_custom_tables_specialised_and_generic(O, FE, lib, c4fm_1s, nt):
    """Random asymmetric tables: the specialised kernels (immediates) and the generic ones (LDS) give the oracle's bits"""
    rng = np.random.default_rng(500 + nt[0])
    dt, ct = rand_taps(rng, nt[0]), rand_taps(rng, nt[1])
    iq = c4fm_1s[0][:140000]
    ok = dict(decim_taps=dt, chan_taps=ct)
    all_paths(O, FE, lib, iq, ok, dict(ok, specialize=lib.SPECIALIZE_REQUIRE), lib.VARIANT_SPECIALIZED)
    all_paths(O, FE, lib, iq, ok, dict(ok, specialize=lib.SPECIALIZE_OFF), lib.VARIANT_GENERIC)


def test_default_numbers_builtin_and_forced_specialisation(O, FE, lib, c4fm_1s):
    """The build's own numbers: the library's code object, and the hipRTC product of the same source (the two must agree
    with the oracle, hence with each other)"""
    iq = c4fm_1s[0][:140000]
    all_paths(O, FE, lib, iq, {}, {}, lib.VARIANT_BUILTIN)
    all_paths(O, FE, lib, iq, {}, dict(specialize=lib.SPECIALIZE_FORCE), lib.VARIANT_SPECIALIZED)
    # naming the defaults explicitly is still "the build's own numbers"
    spec = O.load_spec()
    # (fma(b, s, -1) exactly: b * s and the sum are exact in double, rounded once)
    lut = np.array([np.float32(np.float64(b) * np.float64(np.float32(spec["u8_scale"])) - 1.0) for b in range(256)], dtype=np.float32)
    fe = FE(fm_deviation_hz=5000, fm_sample_rate_hz=48000, u8_scale=spec["u8_scale"], u8_offset=-1.0, fm_gain=spec["fm_gain"], u8_lut=lut)
    assert fe.kernel_variant == lib.VARIANT_BUILTIN


@pytest.mark.parametrize("mode", ["specialised", "generic"])
def test_runtime_fm_deviation_and_u8_constants(O, FE, lib, c4fm_1s, mode):
    """FmDemod::new(deviation, rate) and the u8 conversion as run-time arguments (ABI 4)"""
    sp = lib.SPECIALIZE_REQUIRE if mode == "specialised" else lib.SPECIALIZE_OFF
    var = lib.VARIANT_SPECIALIZED if mode == "specialised" else lib.VARIANT_GENERIC
    iq = c4fm_1s[0][:100000]
    # another deviation (the C4FM levels then read +-0.45 / +-0.15: the slicer's thresholds follow the sync word)
    kw = dict(fm_deviation_hz=4000)
    all_paths(O, FE, lib, iq, kw, dict(kw, specialize=sp), var)
    # an explicit scale (demod_fm's own constant, once someone dumps it) wins over deviation / rate
    g = float(np.nextafter(np.float32(O.load_spec()["fm_gain"]), np.float32(2)))
    all_paths(O, FE, lib, iq, dict(fm_gain=g), dict(fm_gain=g, fm_deviation_hz=1234, specialize=sp), var)
    # (b - 127.5) / 127.5 style and (b - 127) / 128 style tables as scale / offset
    kw = dict(u8_scale=float(np.float32(1 / 128.0)), u8_offset=float(np.float32(-127 / 128.0)))
    all_paths(O, FE, lib, iq, kw, dict(kw, specialize=sp), var)


@pytest.mark.parametrize("mode", ["specialised", "generic"])
def test_u8_lookup_table(O, FE, lib, c4fm_1s, mode):
    """rtlsdr_iq::IQ as a TABLE (src/demod.rs:83): an affine table runs as arithmetic, any other one is looked up in LDS"""
    from p25rx_amd import c4fm
    sp = lib.SPECIALIZE_REQUIRE if mode == "specialised" else lib.SPECIALIZE_OFF
    var = lib.VARIANT_SPECIALIZED if mode == "specialised" else lib.VARIANT_GENERIC
    iq = c4fm_1s[0][:100000]
    # correctly rounded (b - 127.5) / 127.5: NOT an fma of the byte for every b -> the LDS table
    lut = ((np.arange(256, dtype=np.float64) - 127.5) / 127.5).astype(np.float32)
    kw = dict(u8_lut=lut)
    all_paths(O, FE, lib, iq, kw, dict(kw, specialize=sp), var)
    # a companded table (nothing like a line), on bytes that cover the whole range
    lut2 = np.tanh((np.arange(256) - 127.5) / 70.0).astype(np.float32)
    rng = np.random.default_rng(9)
    u8 = c4fm.to_u8(iq)
    u8[::7] = rng.integers(0, 256, size=len(u8[::7]), dtype=np.uint8)
    all_paths(O, FE, lib, iq, dict(u8_lut=lut2), dict(u8_lut=lut2, specialize=sp), var, u8=u8)


def test_cache_is_used_and_survives_damage(FE, lib, tmp_path, monkeypatch):
    """The code object of a set of numbers is compiled once, found again, and rebuilt if the cached file is damaged"""
    import time
    d = tmp_path / "cache"
    monkeypatch.setenv("P25FE_CACHE_DIR", str(d))
    monkeypatch.delenv("P25FE_SPEC_DIR", raising=False)
    rng = np.random.default_rng(77)
    kw = dict(decim_taps=rand_taps(rng, 31), chan_taps=rand_taps(rng, 41), specialize=lib.SPECIALIZE_REQUIRE)
    t0 = time.perf_counter()
    fe = FE(**kw)
    t_cold = time.perf_counter() - t0
    files = sorted(os.listdir(d))
    assert fe.kernel_variant == lib.VARIANT_SPECIALIZED and len(files) == 1 and files[0].endswith(".hsaco")
    t0 = time.perf_counter()
    fe2 = FE(**kw)
    t_warm = time.perf_counter() - t0
    assert fe2.kernel_variant == lib.VARIANT_SPECIALIZED and t_warm < t_cold / 3, (t_cold, t_warm)
    # an ahead-of-time directory ($P25FE_SPEC_DIR, the `make SPEC=` deployment) is looked in first
    aot = tmp_path / "aot"
    name = lib.specialize(lib.make_config(**kw), str(aot))
    os.unlink(d / files[0])
    monkeypatch.setenv("P25FE_SPEC_DIR", str(aot))
    assert FE(**kw).kernel_variant == lib.VARIANT_SPECIALIZED and os.listdir(d) == [] and os.path.exists(name)
    monkeypatch.delenv("P25FE_SPEC_DIR")
    # damage: a truncated ELF -> recompiled and replaced
    with open(d / files[0], "wb") as f:
        f.write(open(name, "rb").read()[:4096])
    fe3 = FE(**kw)
    assert fe3.kernel_variant == lib.VARIANT_SPECIALIZED and os.path.getsize(d / files[0]) > 4096
    x = np.zeros(16384, dtype=np.complex64)
    assert len(fe3.demod_cf32(x)) == 3276


_AUTO_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from p25rx_amd import _lib, c4fm
from p25rx_amd.frontend import FrontEnd
rng = np.random.default_rng(%(seed)d)
taps = lambda n: (rng.standard_normal(n) * np.hanning(n + 2)[1:-1] / 6).astype(np.float32).tolist()
dt, ct = taps(31), taps(41)
iq = c4fm.synth(0.5, seed=9, snr_db=25.0)[0]
ocfg = O.make_config(O.load_spec(), dt, ct)
ref = O.Demod(ocfg).feed_cf32(iq)
dref = O.Recv(ocfg).feed(ref)[0]
fe = FrontEnd(decim_taps=dt, chan_taps=ct)                         # specialize = AUTO, P25FE_JIT unset: the production default
got = fe.demod_cf32(iq)
fe.reset()
dib = np.concatenate([fe.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
print(json.dumps({"variant": fe.kernel_variant, "bb": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))),
                  "dib": bool(np.array_equal(dib, dref)), "log": _lib.specialize_log()[-1500:]}))
'''


def _auto_run(env_extra, seed=41):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    env.pop("P25FE_JIT", None)                                      # conftest.py switches AUTO off for the suite's own process
    env.pop("P25FE_SPEC_DIR", None)
    r = subprocess.run([sys.executable, "-c", _AUTO_WORKER % {"root": root, "seed": seed}], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


@pytest.mark.timeout(600)
def test_auto_mode_is_the_production_default_and_degrades_loudly(lib, tmp_path):
    """The suite runs with P25FE_JIT=0 (tests/conftest.py); the PRODUCTION default is specialize = AUTO with nothing in the
    environment: cache lookup, then hipRTC, then the generic kernels.  Each leg in a process of its own: (a) cold cache ->
    compiled, stored, specialised kernels; (b) warm cache -> found; (c) hipRTC missing and nothing cached -> the generic kernels,
    announced on stderr; (d) a cache directory that cannot be written -> compiled and used from memory.  Same bits as the
    oracle on every leg."""
    cache = tmp_path / "cache"
    a, _ = _auto_run({"P25FE_CACHE_DIR": str(cache)})
    assert a["variant"] == lib.VARIANT_SPECIALIZED and a["bb"] and a["dib"] and len(os.listdir(cache)) == 1
    b, _ = _auto_run({"P25FE_CACHE_DIR": str(cache), "P25FE_HIPRTC": "/nonexistent/libhiprtc.so"})
    # (without hipRTC the versioned key cannot even be formed: only ahead-of-time objects are found -- this one is not)
    assert b["variant"] == lib.VARIANT_GENERIC and b["bb"] and b["dib"]
    b2, _ = _auto_run({"P25FE_CACHE_DIR": str(cache)})
    assert b2["variant"] == lib.VARIANT_SPECIALIZED and b2["bb"] and b2["dib"] and len(os.listdir(cache)) == 1
    c, err = _auto_run({"P25FE_CACHE_DIR": str(tmp_path / "empty"), "P25FE_HIPRTC": "/nonexistent/libhiprtc.so"})
    assert c["variant"] == lib.VARIANT_GENERIC and c["bb"] and c["dib"] and "running the GENERIC kernels" in err
    ro = tmp_path / "readonly"
    ro.mkdir(mode=0o500)
    d, _ = _auto_run({"P25FE_CACHE_DIR": str(ro)}, seed=42)
    assert d["bb"] and d["dib"]
    if os.geteuid() != 0:                                           # (root writes through a 0500 directory)
        assert d["variant"] == lib.VARIANT_SPECIALIZED and "used from memory" in d["log"] and os.listdir(ro) == []


def _abi5_cases(spec):
    rng = np.random.default_rng(2024)
    r = lambda n: (rng.standard_normal(n) / 8).astype(np.float32).tolist()
    return {
        # name: (config keywords, which kernels may run it)
        "rc41_phase2": (dict(avg_taps=spec["rc_avg_taps"], decim_phase=2), ("spec", "generic")),        # LDS window, raised cosine
        "rand21_phase0": (dict(avg_taps=r(21), decim_phase=0), ("spec", "generic")),
        "boxcar7_phase3": (dict(avg_taps=[float(np.float32(1.0 / 7.0))] * 7, decim_phase=3), ("spec", "generic")),   # registers, uniform
        "rand11_phase1": (dict(avg_taps=r(11), decim_phase=1), ("spec", "generic")),                    # registers, fma chain
        "one_tap": (dict(avg_taps=[1.0]), ("spec", "generic")),
        "rand64_long_tables": (dict(avg_taps=r(64), decim_taps=rand_taps(rng, 64), chan_taps=rand_taps(rng, 64), decim_phase=2),
                               ("spec", "generic")),                                                # 160-output segment halo
        "default_avg_phase1": (dict(decim_phase=1), ("builtin",)),                                     # the phase is the host's: built-in kernels
        "default_avg_as_a_table": (dict(avg_taps=spec["avg_taps"], decim_phase=4), ("builtin",)),
    }


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", ["rc41_phase2", "rand21_phase0", "boxcar7_phase3", "rand11_phase1", "one_tap", "rand64_long_tables",
                                  "default_avg_phase1", "default_avg_as_a_table"])
def test_abi5_post_discriminator_filter_and_decimator_phase(O, FE, lib, case, c4fm_1s):
    """ABI 5: MovingAverage::new(10) as a table (src/demod.rs:52, 114) and Decimator::new(5)'s phase (src/demod.rs:50, 87-90) --
    the last two constructor numbers -- through every kernel shape (linear, planar, one-launch chunk), cf32 and u8, in the
    kernels specialised for the numbers AND in the generic ones, bit for bit against the oracle configured the same way.
    Ten equal taps of 0.1 and phase 4 are the build's own numbers and run the built-in kernels."""
    spec = O.load_spec()
    kw, runs = _abi5_cases(spec)[case]
    iq = c4fm_1s[0][:120000]
    for run in runs:
        fe_kw = dict(kw)
        if run == "spec":
            fe_kw["specialize"] = lib.SPECIALIZE_REQUIRE
            want = lib.VARIANT_SPECIALIZED
        elif run == "generic":
            fe_kw["specialize"] = lib.SPECIALIZE_OFF
            want = lib.VARIANT_GENERIC
        else:
            want = lib.VARIANT_BUILTIN
        fe = all_paths(O, FE, lib, iq, kw, fe_kw, want)
        # a device range that starts mid-stream, its history in memory: the output count follows the handle's phase
        import torch
        t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
        a0, h = 40003 // 2 * 2, fe.shard_halo()
        ref = O.Demod(O.make_config(spec, **kw)).feed_cf32(iq)
        fe.reset()
        bb, nb = fe.demod_dev(t[a0 - h:], n_hist=h, abs0=a0, offset=h)
        assert nb == fe.n_baseband(a0, len(iq) - a0) and np.array_equal(bits(bb[0, :nb].cpu().numpy()), bits(ref[len(ref) - nb:]))


def test_abi5_time_shards_with_a_decimator_phase_and_a_long_filter(O, FE, lib):
    """Time shards with the ABI 5 numbers: the shards' baseband ranges follow the handle's decimator phase (p25fe_n_baseband_h) and
    the halo (2 560 samples) covers a 41-tap post-discriminator filter behind the 41-tap channel filter: == one pass == oracle."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    spec = O.load_spec()
    kw = dict(avg_taps=spec["rc_avg_taps"], decim_phase=1)
    iq = c4fm.synth(2.0, seed=34, snr_db=22.0, frame_dibits=900)[0]
    ocfg = O.make_config(spec, **kw)
    ref = O.Recv(ocfg).feed(O.Demod(ocfg).feed_cf32(iq))[0]
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    for sp in (lib.SPECIALIZE_REQUIRE, lib.SPECIALIZE_OFF):
        fes = [FE(specialize=sp, **kw) for _ in range(3)]
        halo = fes[0].shard_halo()
        assert halo == 2560
        cuts = [0, 150000, 150008 + 8000, len(iq) // 8 * 8]
        summ, bb0, bbn = [], [], []
        for r in range(3):
            a, b = cuts[r], cuts[r + 1]
            h = min(a, halo)
            fes[r].shard_pass1_main(t[a - h:b], offset=h, n_hist=h, abs0=a)
            res = fes[r].shard_pass1_finish(t[a - h:b], offset=h, n_hist=h, abs0=a)
            summ.append(parse_results(res)[0])
            bb0.append(fes[r].n_baseband(0, a))
            bbn.append(fes[r].n_baseband(a, b - a))
        summ_t = torch.from_numpy(np.frombuffer(np.array(summ).tobytes(), dtype=np.uint8).copy()).view(3, -1).cuda()
        d_bb0, d_bbn = torch.tensor(bb0, dtype=torch.int64, device="cuda"), torch.tensor(bbn, dtype=torch.int64, device="cuda")
        out = []
        for r in range(3):
            dib, res, _, off = fes[r].shard_pass2_dev(summ_t, d_bb0, d_bbn, r, bbn[r])
            out.append(dib[0, :int(parse_results(res)[0]["n_dibits"])].cpu().numpy())
        got = np.concatenate(out)
        nref = O.Recv(ocfg).feed(O.Demod(ocfg).feed_cf32(iq[:cuts[-1]]))[0]
        assert np.array_equal(got, nref) and int(off[-1]) == len(nref)
