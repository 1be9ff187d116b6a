"""CPU tests that pin the oracle (oracle/p25fe_oracle.c).

The reference has no golden vectors for this path (parity unpinned), so the
oracle is pinned against an independent fp64 numpy/scipy model, physical known answers, its
own streaming invariances and the known symbols of the seeded modulator.
"""
import json
import os

import numpy as np
import pytest
from scipy import signal

from oracle import oracle as O
from p25rx_amd import c4fm

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def fp64_model(iq, spec):
    """Independent model: scipy lfilter / np.angle in fp64."""
    h1 = np.array(spec["decim_taps"], dtype=np.float64)
    h2 = np.array(spec["chan_taps"], dtype=np.float64)
    x = iq.astype(np.complex128)
    d = signal.lfilter(h1, 1.0, x)[spec["decim"] - 1::spec["decim"]]
    y = signal.lfilter(h2, 1.0, d)
    prev = np.concatenate([[0.0], y[:-1]])
    fm = np.angle(y * np.conj(prev)) * (48000.0 / (2 * np.pi * 5000.0))
    bb = signal.lfilter(np.ones(10) / 10.0, 1.0, fm)
    return y, fm, bb


def test_fp64_cross_check(spec, c4fm_1s):
    iq, _, _ = c4fm_1s
    iq = iq[:60000]
    ch, fm, bb = O.Demod().feed_cf32_stages(iq)
    y, fm64, bb64 = fp64_model(iq, spec)
    assert len(bb) == len(bb64) == 12000
    assert np.abs(ch - y).max() < 2e-6
    # skip the first samples where |y| ~ 0 makes the angle ill-conditioned
    assert np.abs(fm[100:] - fm64[100:]).max() < 2e-5
    assert np.abs(bb[100:] - bb64[100:]).max() < 1e-5


def test_chunk_lengths_and_invariance(c4fm_1s):
    """16384-sample chunks give 3276/3277 outputs (src/demod.rs:87-90) and any chunking is bit-identical."""
    iq, _, _ = c4fm_1s
    one = O.Demod().feed_cf32(iq)
    d = O.Demod()
    lens, parts = [], []
    for off in range(0, len(iq), 16384):
        p = d.feed_cf32(iq[off:off + 16384])
        lens.append(len(p))
        parts.append(p)
    assert lens[:5] == [3276, 3277, 3277, 3277, 3277][:5] or set(lens[:-1]) == {3276, 3277}
    assert set(lens[:-1]) == {3276, 3277}
    assert np.array_equal(np.concatenate(parts), one)
    rng = np.random.default_rng(3)
    d = O.Demod()
    parts, off = [], 0
    while off < len(iq):
        n = int(rng.integers(1, 5000))
        parts.append(d.feed_cf32(iq[off:off + n]))
        off += n
    assert np.array_equal(np.concatenate(parts), one)


def test_u8_lut_path(spec):
    """u8 pairs -> LUT (src/demod.rs:74-84): byte 2i is I, byte 2i+1 is Q; value fma(b, 2/255, -1)."""
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, size=2 * 20000, dtype=np.uint8)
    lut = (raw.astype(np.float64) * np.float64(np.float32(spec["u8_scale"])) - 1.0).astype(np.float32)
    iq = (lut[0::2] + 1j * lut[1::2]).astype(np.complex64)
    a = O.Demod().feed_u8(raw)
    b = O.Demod().feed_cf32(iq)
    assert np.array_equal(a, b)


def test_kat_dc_and_tone(spec):
    n = 48000
    t = np.arange(n) / 240000.0
    dc = (0.5 * np.ones(n)).astype(np.complex64)
    bb = O.Demod().feed_cf32(dc)
    assert np.all(bb[80:] == 0.0)
    for f in (1800.0, -600.0, 2500.0):
        tone = (0.5 * np.exp(2j * np.pi * f * t)).astype(np.complex64)
        bb = O.Demod().feed_cf32(tone)
        assert np.abs(bb[80:] - f / 5000.0).max() < 5e-6


def test_kat_impulse_reads_back_taps(spec):
    """An impulse at input n0 = 5k+4 puts decim tap 5j into output k+j; channel taps follow."""
    x = np.zeros(2000, dtype=np.complex64)
    x[4] = 1.0
    ch, _, _ = O.Demod().feed_cf32_stages(x)
    h1 = np.array(spec["decim_taps"], dtype=np.float32)[0::5]
    h2 = np.array(spec["chan_taps"], dtype=np.float32)
    ref = np.convolve(h1.astype(np.float64), h2.astype(np.float64))
    assert np.abs(ch[:len(ref)].real - ref).max() < 1e-7
    assert np.all(ch.imag == 0)


def test_power_dbm_formula():
    """demod::power_dbm (src/demod.rs:123-134): constant envelope A -> 30 + 10 log10(A^2)."""
    for a in (1.0, 0.5, 0.01):
        x = (a * np.exp(1j * np.linspace(0, 50, 3277))).astype(np.complex64)
        assert abs(O.power_dbm(x) - (30 + 20 * np.log10(a))) < 1e-3


def test_atan2_against_libm(spec):
    rng = np.random.default_rng(11)
    xs = rng.standard_normal(20000).astype(np.float32)
    ys = rng.standard_normal(20000).astype(np.float32)
    cfg = O.make_config()
    got = np.array([O.atan2f(float(y), float(x), cfg) for x, y in zip(xs[:4000], ys[:4000])])
    ref = np.arctan2(ys[:4000].astype(np.float64), xs[:4000].astype(np.float64))
    assert np.abs(got - ref).max() < 4e-7
    assert O.atan2f(0.0, 0.0, cfg) == 0.0
    assert abs(O.atan2f(0.0, -1.0, cfg) - np.pi) < 1e-6
    assert abs(O.atan2f(1.0, 0.0, cfg) - np.pi / 2) < 1e-6
    assert abs(O.atan2f(-1.0, 0.0, cfg) + np.pi / 2) < 1e-6


def test_recv_decodes_ground_truth(c4fm_1s):
    iq, truth, _ = c4fm_1s
    bb = O.Demod().feed_cf32(iq)
    dib, spos, sdib = O.Recv().feed(bb)
    assert len(spos) == 6 and np.all(np.diff(spos) == 8640)
    assert list(sdib[:3]) == [0, 864, 1728]
    n = min(len(dib), len(truth) - 24)
    assert n > 4700
    assert np.array_equal(dib[:n], truth[24:24 + n])


def test_recv_chunk_invariance_and_resync(c4fm_1s):
    iq, truth, _ = c4fm_1s
    bb = O.Demod().feed_cf32(iq)
    ref = O.Recv().feed(bb)
    r = O.Recv()
    rng = np.random.default_rng(9)
    outs, off = [], 0
    while off < len(bb):
        n = int(rng.integers(1, 3000))
        outs.append(r.feed(bb[off:off + n]))
        off += n
    assert np.array_equal(np.concatenate([o[0] for o in outs]), ref[0])
    assert np.array_equal(np.concatenate([o[1] for o in outs]), ref[1])
    assert np.array_equal(np.concatenate([o[2] for o in outs]), ref[2])
    # resync (src/recv.rs:136,179) before sample 20000: no dibits until the next sync word (26222)
    r = O.Recv()
    a = r.feed(bb[:20000])
    r.resync()
    b = r.feed(bb[20000:])
    assert b[1][0] == 26222
    assert len(a[0]) == len(ref[0][ref_instants(ref, 20000)])
    assert np.array_equal(b[0], ref[0][len(ref[0]) - len(b[0]):])
    assert len(b[0]) == (len(bb) - 1 - 26222) // 10


def ref_instants(ref, upto):
    """Slice selecting the dibits of `ref` whose instants are < upto (first sync at ref[1][0], stride 10)."""
    s0 = int(ref[1][0])
    return slice(0, (upto - 1 - s0) // 10)


def test_recv_with_offsets_and_noise():
    """Frequency offset (DC after FM) and 15 dB SNR still decode: thresholds track the sync levels."""
    iq, truth, _ = c4fm.synth(0.5, seed=4, snr_db=15.0, freq_offset_hz=300.0, timing_offset=17)
    bb = O.Demod().feed_cf32(iq)
    dib, spos, _ = O.Recv().feed(bb)
    assert len(spos) >= 2
    n = min(len(dib), len(truth) - 24)
    err = np.count_nonzero(dib[:n] != truth[24:24 + n])
    assert err <= n * 0.01


def test_golden_fixture_matches():
    """Self-generated golden (tests/golden/gen_golden.py): pins the oracle across rounds."""
    g = np.load(os.path.join(GOLDEN, "c4fm_seed7_u8.npz"))
    d = O.Demod()
    bb = np.concatenate([d.feed_u8(g["iq_u8"][o:o + 32768]) for o in range(0, len(g["iq_u8"]), 32768)])
    assert np.array_equal(bb.view(np.uint32), g["bb_bits"])
    dib, spos, sdib = O.Recv().feed(bb)
    assert np.array_equal(dib, g["dibits"])
    assert np.array_equal(spos, g["sync_pos"])
    with open(os.path.join(GOLDEN, "spec.json")) as f:
        assert json.load(f)["t1"] == 31


def test_golden_dense_planes_fixture_matches(spec):
    """Self-generated golden of the receiver on the densest scene (tests/golden/gen_golden_dense.py: 320 detections per tile, three
    clocks, with and without lock drops): the oracle, fed in ragged chunks where the clock streams, and the independent batch model
    reproduce the committed vectors."""
    import spec_model
    g = np.load(os.path.join(GOLDEN, "dense_planes.npz"))
    bb = g["bb_bits"].view(np.float32)
    for mode in (0, 1, 2):
        for tag, drops in (("", []), ("_drops", [int(x) for x in g["drops"]])):
            want = [g["dibits_m%d%s" % (mode, tag)], g["sync_pos_m%d%s" % (mode, tag)], g["sync_dibit_m%d%s" % (mode, tag)]]
            got = O.recv_range(bb, O.make_config(symbol_clock=mode), drops)
            m = spec_model.Model(spec)
            ref = m.receive(bb, drops) if mode == 0 else m.receive_tracking(bb, spec, drops, reslice=(mode == 2))
            for k in range(3):
                assert np.array_equal(got[k], want[k].astype(got[k].dtype)), (mode, tag, k)
                assert np.array_equal(np.asarray(ref[k]).astype(want[k].dtype), want[k]), (mode, tag, k, "model")
            if mode < 2:                                            # streaming: any chunking gives the same output
                r = O.Recv(O.make_config(symbol_clock=mode))
                outs, o = [], 0
                for c in sorted(set(drops + [333, 3276, 3277, 9000, len(bb)])):
                    outs.append(r.feed(bb[o:c]))
                    if c in drops:
                        r.resync()
                    o = c
                assert np.array_equal(np.concatenate([x[0] for x in outs]), want[0]) and np.array_equal(np.concatenate([x[1] for x in outs]), want[1])


def test_predecim_fp64_and_chunking(spec):
    """Stage 0 (config 3, no reference counterpart): 10:1 decimating FIR vs scipy fp64; chunk invariance."""
    rng = np.random.default_rng(0)
    x = ((rng.standard_normal(60001) + 1j * rng.standard_normal(60001)) * 0.3).astype(np.complex64)
    y = O.PreDecim().feed(x)
    ref = signal.lfilter(np.array(spec["pre_taps"], dtype=np.float64), 1.0, x.astype(np.complex128))[9::10]
    assert len(y) == len(ref) == 6000
    assert np.abs(y - ref).max() < 3e-7
    p = O.PreDecim()
    parts = [p.feed(x[o:o + 777]) for o in range(0, len(x), 777)]
    assert np.array_equal(np.concatenate(parts), y)


def test_nid_code_and_decode(spec):
    """SPEC 3.9: BCH(63,16,23) NID after each frame sync.  The generator polynomial is verified in tools/gen_spec.py
    (degree 47, divides x^63 + 1, minimum weight 23); here: encode -> modulate -> demodulate -> decode, and the
    error-correction radius (11 corrected, 12 rejected)."""
    nidf = lambda f: (0x293 + f, (3 * f) & 15)
    iq, truth, _ = c4fm.synth(1.0, seed=3, snr_db=20.0, nid=nidf)
    dib, spos, sdib = O.Recv().feed(O.Demod().feed_cf32(iq))
    nids = O.nid_decode(dib, sdib, spos)
    assert len(nids) == 6 and np.all(nids["valid"] == 1) and np.all(nids["n_errors"] == 0)
    assert [int(x) for x in nids["nac"]] == [0x293 + f for f in range(6)]
    assert [int(x) for x in nids["duid"]] == [(3 * f) & 15 for f in range(6)]
    assert np.array_equal(nids["sync_pos"], spos)
    assert int(nids["raw"][0]) == c4fm.nid_word(0x293, 0)
    D = int(sdib[2])
    bad = dib.copy()
    flips = [0, 2, 4, 6, 8, 13, 15, 17, 19, 21, 23]                 # 11 single-bit errors, none on the status symbol (D + 11)
    for j in flips:
        bad[D + j] ^= 1
    r = O.nid_decode(bad, sdib[2:3])[0]
    assert r["valid"] == 1 and r["n_errors"] == 11 and r["nac"] == 0x295 and r["duid"] == 6
    bad[D + 25] ^= 2
    r = O.nid_decode(bad, sdib[2:3])[0]
    assert r["valid"] == 0 and r["n_errors"] >= 12
    bad2 = dib.copy()
    bad2[D + 11] ^= 3                                                # the status symbol is not part of the NID
    assert O.nid_decode(bad2, sdib[2:3])[0]["n_errors"] == 0
    assert O.nid_decode(dib[:D + 20], sdib[2:3])[0]["valid"] == -1   # stream ends inside the NID


def test_channeliser_oracle(spec):
    """SPEC 3.11 (next row): the plain mix / filter / decimate restatement against numpy in fp64,
    its channel 0 against the pre-decimator of 3.0, chunk invariance through (n_hist, abs0), and a tone KAT: a tone
    d Hz above raster slot c comes out of channel c at d Hz with the prototype's gain, and is absent 40 slots away."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 12000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.3
    y = O.channelise(x)
    assert y.shape == (192, n // 10)
    h = np.array(spec["pre_taps"], dtype=np.float64)
    k = np.arange(n)
    for c in (0, 1, 37, 95, 96, 191):
        z = np.convolve(x.astype(np.complex128) * np.exp(-2j * np.pi * ((c * k) % 192) / 192), h)[:n][9::10]
        assert np.abs(z - y[c]).max() < 5e-7
    assert np.abs(y[0] - O.PreDecim().feed(x)).max() < 1e-6          # 3.0 is channel 0 (fp32 vs fp64 accumulation)
    for off in (1000, 1234, 5557):
        y2 = O.channelise(x[off - 79:], n_hist=79, abs0=off)
        first = len([m for m in range(n // 10) if 10 * m + 9 < off])
        assert np.array_equal(y2, y[:, first:])
    c, d = 53, 2000.0
    t = np.arange(48000) / 2.4e6
    tone = (0.5 * np.exp(2j * np.pi * (c * 12500.0 + d) * t)).astype(np.complex64)
    yt = O.channelise(tone)
    g = abs(np.sum(h * np.exp(-2j * np.pi * d / 2.4e6 * np.arange(len(h)))))
    seg = yt[c, 100:]
    assert abs(np.abs(seg).mean() - 0.5 * g) < 1e-4
    f_est = np.angle(np.sum(seg[1:] * np.conj(seg[:-1]))) * 240000 / (2 * np.pi)
    assert abs(f_est - d) < 1.0
    assert np.abs(yt[(c + 40) % 192, 100:]).max() < 1e-3             # 500 kHz away: stop band of the prototype


# ---- SPEC 3.8b: tracking symbol clock (north_star's "symbol-clock interpolator") -----------------------------------
def frames_ok(dib, sdib, truth, frame):
    """Per sync event k: the dibits it governs (up to the next event) against the modulator's symbols that follow
    sync word k (its own 24 dibits are not emitted: the detection is decided at the word's last symbol)."""
    ok = []
    for k in range(len(sdib) - 1):
        got = dib[int(sdib[k]):int(sdib[k + 1])]
        want = truth[frame * k + 24:frame * (k + 1) + 24]
        ok.append(len(got) == len(want) and np.array_equal(got, want))
    return ok


def test_tracking_clock_equals_fixed_stride_on_a_nominal_clock(c4fm_1s):
    """Sync words exactly 8640 samples apart give the period 8640 / 864 = 10 / 1: phase 0 of the interpolator is the
    identity row, so mode 1 emits the same dibits and events as the reference-shaped fixed stride (2 samples later)."""
    iq, truth, _ = c4fm_1s
    bb = O.Demod().feed_cf32(iq)
    ref = O.Recv().feed(bb)
    trk = O.Recv(O.make_config(symbol_clock=1)).feed(bb)
    n = len(trk[0])
    assert len(ref[0]) - 1 <= n <= len(ref[0]) and np.array_equal(trk[0], ref[0][:n])     # the last instant may wait for its lookahead
    assert np.array_equal(trk[1], ref[1]) and np.array_equal(trk[2], ref[2])


@pytest.mark.parametrize("ppm,frame", [(100.0, 3000), (-100.0, 3000), (35.0, 8640)])
def test_tracking_clock_follows_a_sample_clock_error(ppm, frame):
    """+-100 ppm with 0.625 s between sync words, 35 ppm with 1.8 s: the fixed stride walks 3 samples off the eye inside a
    frame and loses symbols (the integrate-and-dump eye closes at ~1.7 samples for a +3 -> -3 transition); with the
    period taken from the previous sync-to-sync interval every frame after the first decodes to the modulator's dibits.
    (The symbol count of an interval is round(interval / 10): unambiguous while ppm x interval + the +-1 sample jitter
    of two detections stays below 5 samples -- the limit of ANY scheme that infers the count from the gap alone,
    interpolation between consecutive sync words included; 100 ppm x 1.8 s = 8.6 samples is beyond it.)"""
    n_frames = 5
    iq, truth, _ = c4fm.synth((n_frames - 1) * frame / 4800.0 + 0.3, seed=21, snr_db=30.0, frame_dibits=frame, clock_ppm=ppm)
    bb = O.Demod().feed_cf32(iq)
    fix = O.Recv().feed(bb)
    trk = O.Recv(O.make_config(symbol_clock=1)).feed(bb)
    assert len(trk[1]) == len(fix[1]) == n_frames and np.array_equal(trk[1], fix[1])
    d = np.diff(trk[1])
    assert np.all(np.abs(d - frame * 10 * (1 + ppm * 1e-6)) <= 1.5)           # the intervals the period estimates come from
    ok_fix, ok_trk = frames_ok(*fix[::2], truth, frame), frames_ok(*trk[::2], truth, frame)
    assert not any(ok_fix)                                                     # every frame of the fixed stride is damaged
    assert not ok_trk[0] and all(ok_trk[1:])                                   # first frame: no estimate yet; then locked
    # chunk invariance holds in mode 1 too (any chunking, state carried)
    r = O.Recv(O.make_config(symbol_clock=1))
    rng = np.random.default_rng(5)
    outs, off = [], 0
    while off < len(bb):
        n = int(rng.integers(1, 5000))
        outs.append(r.feed(bb[off:off + n]))
        off += n
    for q in range(3):
        assert np.array_equal(np.concatenate([o[q] for o in outs]), trk[q])


@pytest.mark.parametrize("seed", range(24))
def test_receiver_against_an_independent_batch_model(spec, seed):
    """The oracle's streaming receiver (per-sample feed objects, state across chunks, resync between chunks) against
    tests/spec_model.py -- SPEC 3.7 / 3.8 restated from the text as a batch computation in numpy float32: dibits, sync
    positions and sync dibit indices must be identical over random SNRs (down to false and missed sync words), frame
    lengths (down to back-to-back sync words), offsets, chunkings and lock drops."""
    import spec_model
    from p25rx_amd import c4fm
    rng = np.random.default_rng(1000 + seed)
    snr = float(rng.choice([30.0, 12.0, 6.0, 3.0, 0.0]))
    frame = int(rng.choice([24, 30, 48, 100, 864]))
    iq, _, _ = c4fm.synth(float(rng.choice([0.1, 0.3, 0.6])), seed=seed, snr_db=snr, frame_dibits=frame,
                          freq_offset_hz=float(rng.choice([0.0, 300.0, -800.0])), timing_offset=int(rng.integers(0, 50)),
                          amplitude=float(rng.choice([0.5, 0.05])))
    bb = O.Demod().feed_cf32(iq)
    drops = sorted(set(int(x) for x in rng.integers(0, len(bb) + 3, size=int(rng.choice([0, 0, 2, 15])))))
    r = O.Recv()
    outs, o = [], 0
    cuts = sorted(set(drops + [int(x) for x in rng.integers(0, len(bb), size=6)] + [len(bb)]))
    for q in cuts:                                               # random chunking; resync() exactly at the drop indices
        q = min(q, len(bb))
        outs.append(r.feed(bb[o:q]))
        if q in drops:
            r.resync()
        o = q
    got = [np.concatenate([x[k] for x in outs]) for k in range(3)]
    m = spec_model.Model(spec)
    ref = m.receive(bb, drops)
    same = all(len(got[k]) == len(ref[k]) and np.array_equal(got[k], ref[k].astype(got[k].dtype)) for k in range(3))
    # (the model's energy is a double-rounded fma: a candidate within 1e-6 of a threshold may legitimately fall either way)
    assert same or m.margins() < 1e-6, (seed, snr, frame, len(got[0]), len(ref[0]), len(got[1]), len(ref[1]))
    if snr >= 12.0 and frame >= 48:
        assert len(ref[1]) >= 2 and len(ref[0]) > 50


@pytest.mark.parametrize("fmt", ["cf32", "u8"])
def test_demod_against_an_independent_batch_model(spec, c4fm_1s, fmt):
    """Stages 3.1 - 3.5 of the oracle (streaming objects, ragged chunks) against tests/spec_model.demod -- the same SPEC
    text as whole-array numpy float32 operations with the fused multiply-adds evaluated through float64: equal BIT FOR BIT
    (a double rounding could move a last bit about once in 2^29 operations; none does on these captures)."""
    import spec_model
    from p25rx_amd import c4fm
    iq = c4fm_1s[0][:60000]
    d = O.Demod()
    if fmt == "u8":
        raw = c4fm.to_u8(iq)
        got = np.concatenate([d.feed_u8(raw[o:o + 2 * 7001]) for o in range(0, len(raw), 2 * 7001)])
        ref = spec_model.demod(spec, u8=raw)
    else:
        got = np.concatenate([d.feed_cf32(iq[o:o + 7001]) for o in range(0, len(iq), 7001)])
        ref = spec_model.demod(spec, iq=iq)
    assert len(got) == len(ref) == 12000
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # and the receiver on top: the two independent restatements give the same symbols end to end
    m = spec_model.Model(spec)
    a, b = O.Recv().feed(got), m.receive(ref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and len(a[0]) > 1000


@pytest.mark.parametrize("case", ["rc41_phase2", "ramp21_phase0", "boxcar7_phase3", "default_table", "one_tap"])
def test_demod_constructor_numbers_against_the_independent_model(spec, c4fm_1s, case):
    """ABI 5: the last two constructor numbers of DemodTask::new as data -- Decimator::new(5)'s phase (src/demod.rs:50, 87-90) and
    MovingAverage::new(10) as a table (src/demod.rs:52, 114) -- in the oracle (streaming objects, ragged chunks) against the
    whole-array model of the same SPEC text (3.2, 3.5): bit for bit, cf32 and u8; the default table IS the old boxcar."""
    import spec_model
    from p25rx_amd import c4fm
    rng = np.random.default_rng(3)
    kw = {"rc41_phase2": dict(avg_taps=spec["rc_avg_taps"], decim_phase=2),
          "ramp21_phase0": dict(avg_taps=(rng.standard_normal(21) / 8).astype(np.float32).tolist(), decim_phase=0),
          "boxcar7_phase3": dict(avg_taps=[float(np.float32(1.0 / 7.0))] * 7, decim_phase=3),
          "default_table": dict(avg_taps=spec["avg_taps"], decim_phase=4),
          "one_tap": dict(avg_taps=[1.0], decim_phase=1)}[case]
    iq = c4fm_1s[0][:50003]
    raw = c4fm.to_u8(iq)
    d = O.Demod(O.make_config(spec, **kw))
    got = np.concatenate([d.feed_cf32(iq[o:o + 7001]) for o in range(0, len(iq), 7001)])
    ref = spec_model.demod(spec, iq=iq, **kw)
    assert len(got) == len(ref) and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    d8 = O.Demod(O.make_config(spec, **kw))
    got8 = np.concatenate([d8.feed_u8(raw[o:o + 2 * 4099]) for o in range(0, len(raw), 2 * 4099)])
    ref8 = spec_model.demod(spec, u8=raw, **kw)
    assert np.array_equal(got8.view(np.uint32), ref8.view(np.uint32))
    if case == "default_table":
        assert np.array_equal(got.view(np.uint32), O.Demod().feed_cf32(iq).view(np.uint32))
    # the number of outputs follows the phase: input 5 m + phase must exist
    assert len(got) == (len(iq) - kw["decim_phase"] + 4) // 5


@pytest.mark.parametrize("seed", range(16))
def test_tracking_receiver_against_an_independent_batch_model(spec, seed):
    """SPEC 3.8b (the tracking symbol clock: period from sync word to sync word, interpolated instants, lookahead 2) the
    same way: the oracle's streaming receiver in mode 1, fed in random chunks with lock drops, against
    spec_model.Model.receive_tracking over the whole array -- sample-clock errors up to 250 ppm, frames from back-to-back
    sync words to 3 000 dibits."""
    import spec_model
    from p25rx_amd import c4fm
    rng = np.random.default_rng(7000 + seed)
    snr = float(rng.choice([30.0, 12.0, 6.0, 3.0]))
    frame = int(rng.choice([24, 30, 48, 100, 864, 3000]))
    ppm = float(rng.choice([0.0, 40.0, -100.0, 250.0]))
    iq, _, _ = c4fm.synth(float(rng.choice([0.1, 0.3, 0.7, 1.5])), seed=seed, snr_db=snr, frame_dibits=frame,
                          timing_offset=int(rng.integers(0, 50)), clock_ppm=ppm)
    bb = O.Demod().feed_cf32(iq)
    drops = sorted(set(int(x) for x in rng.integers(0, len(bb) + 3, size=int(rng.choice([0, 0, 2, 15])))))
    r = O.Recv(O.make_config(symbol_clock=1))
    outs, o = [], 0
    for q in sorted(set(drops + [int(x) for x in rng.integers(0, len(bb), size=6)] + [len(bb)])):
        q = min(q, len(bb))
        outs.append(r.feed(bb[o:q]))
        if q in drops:
            r.resync()
        o = q
    got = [np.concatenate([x[k] for x in outs]) for k in range(3)]
    ref = spec_model.Model(spec).receive_tracking(bb, spec, drops)
    for k in range(3):
        assert len(got[k]) == len(ref[k]) and np.array_equal(got[k], ref[k].astype(got[k].dtype)), (seed, snr, frame, ppm, k)


@pytest.mark.parametrize("seed", range(12))
def test_reslice_receiver_against_an_independent_batch_model(spec, seed):
    """SPEC 3.8c (symbol_clock = 2, one resident range): the oracle's two passes -- record every detection's backward clock, give a
    detection without a usable one the clock of the interval that STARTS at it, slice again with the table -- against
    spec_model.Model.receive_tracking(reslice=True), which forms the same clocks from its own list of detections; lock drops inside
    the range, clock errors up to 250 ppm."""
    import spec_model
    from p25rx_amd import c4fm
    rng = np.random.default_rng(8100 + seed)
    snr = float(rng.choice([30.0, 12.0, 6.0]))
    frame = int(rng.choice([24, 48, 100, 864, 3000]))
    ppm = float(rng.choice([0.0, 60.0, -150.0, 250.0]))
    iq, _, _ = c4fm.synth(float(rng.choice([0.3, 0.7, 1.5])), seed=seed, snr_db=snr, frame_dibits=frame,
                          timing_offset=int(rng.integers(0, 50)), clock_ppm=ppm)
    bb = O.Demod().feed_cf32(iq)
    drops = sorted(set(int(x) for x in rng.integers(1, len(bb), size=int(rng.choice([0, 0, 2, 9])))))
    got = O.recv_range(bb, O.make_config(symbol_clock=2), drops)
    ref = spec_model.Model(spec).receive_tracking(bb, spec, drops, reslice=True)
    for k in range(3):
        assert len(got[k]) == len(ref[k]) and np.array_equal(got[k], ref[k].astype(got[k].dtype)), (seed, snr, frame, ppm, k)
    # and it is not the streaming rule in disguise: with a clock error and frames long enough to drift, some scene differs from mode 1
    one = O.recv_range(bb, O.make_config(symbol_clock=1), drops)
    assert len(one[1]) == len(got[1]) and np.array_equal(one[1], got[1])          # the same sync words either way


def test_reslice_table_rule():
    """p25o_reslice_table on a hand-made list of detections: usable clocks are kept; a detection without one takes the NEXT one's when that is
    usable, else the nominal clock; 10 / 1 is written 40 / 4; the anchor offset is f N / 4."""
    import ctypes as C
    rec = np.zeros(6, dtype=O.CLK_DTYPE)
    #            d      n    usable  f        what
    rec[0] = (10,     1,    0,      2)      # first of a lock run, successor usable      -> takes rec[1]'s clock, offset 2 * 864
    rec[1] = (34565,  3456, 1,     -1)      # usable                                      -> kept, offset -1 * 864
    rec[2] = (10,     1,    1,      1)      # usable and exactly nominal                  -> 40 / 4, offset 1
    rec[3] = (10,     1,    0,     -2)      # after a lock drop, successor NOT usable     -> 40 / 4, offset -2
    rec[4] = (10,     1,    0,      0)      # again none, successor usable                -> rec[5]'s clock, offset 0
    rec[5] = (34555,  3456, 1,      2)      # usable, last of the range                   -> kept
    out = np.zeros(6, dtype=O.CLK_DTYPE)
    O.lib().p25o_reslice_table(O._ptr(rec), 6, 10, O._ptr(out))
    got = [(int(r["d"]), int(r["n"]), int(r["f"])) for r in out]
    assert got == [(34565, 3456, 2 * 864), (34565, 3456, -864), (40, 4, 1), (40, 4, -2), (34555, 3456, 0), (34555, 3456, 2 * 864)]


def test_reslice_removes_the_first_frame_errors(spec):
    """What 3.8c is for: P25's 0.18 s between sync words, sample clock 150 and 250 ppm off -- the streaming rule (mode 1) loses a few
    symbols at the end of the first frame of the lock run, which has no period yet; the resident rule (mode 2) loses none."""
    from p25rx_amd import c4fm
    for ppm, min_err1 in ((150.0, 1), (250.0, 10), (-150.0, 1)):
        iq, truth, _ = c4fm.synth(3.0, seed=5, snr_db=30.0, frame_dibits=864, clock_ppm=ppm)
        errs = {}
        for mode in (1, 2):
            cfg = O.make_config(symbol_clock=mode)
            d, _, _ = O.recv_range(O.Demod(cfg).feed_cf32(iq), cfg)
            k = min(len(d), len(truth) - 24)
            errs[mode] = int(np.count_nonzero(d[:k] != truth[24:24 + k]))
        assert errs[2] == 0 and errs[1] >= min_err1, (ppm, errs)


def test_nid_against_the_published_generator_polynomial(spec):
    """SPEC 3.9 independently: all 65 536 code words by polynomial division with the published octal generator (not the
    matrix rows of spec.json): minimum weight 23, the spec's systematic rows are among them, and the oracle's decoder agrees
    with a brute-force nearest-code-word search on words with 0 - 14 random bit errors (inside and outside the radius)."""
    import spec_model
    cw = spec_model.nid_codewords()
    w = spec_model._POP16[(cw & np.uint64(0xffff)).astype(np.int64)] + spec_model._POP16[((cw >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)] \
        + spec_model._POP16[((cw >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)] + spec_model._POP16[((cw >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64)]
    assert int(w[1:].min()) == 23 and int(spec["nid_gen_poly"]) == int(spec_model.NID_GEN_OCTAL, 8)
    rows = [int(r) for r in spec["nid_rows"]]
    assert sorted(rows) == sorted(int(cw[1 << k]) for k in range(16))                       # one row per data bit
    rng = np.random.default_rng(9)
    for trial in range(60):
        data = int(rng.integers(0, 65536))
        nerr = int(rng.choice([0, 1, 5, 11, 11, 12, 14]))
        word = int(cw[data])
        for b in rng.choice(63, size=nerr, replace=False):
            word ^= 1 << int(b)
        extra = int(rng.integers(0, 2))                              # the 64th bit is not part of the code word
        raw = (word << 1) | extra
        # as dibits: 32 NID dibits MSB first with the status symbol interleaved after the 11th
        nid_d = [(raw >> (62 - 2 * k)) & 3 for k in range(32)]
        stream = [0] * 5 + nid_d[:11] + [int(rng.integers(0, 4))] + nid_d[11:] + [0] * 3
        r = O.nid_decode(np.array(stream, dtype=np.uint8), np.array([5], dtype=np.uint64))[0]
        k, dist = spec_model.nid_decode(cw, word)
        assert int(r["raw"]) == raw and int(r["n_errors"]) == dist and int(r["valid"]) == (1 if dist <= 11 else 0)
        assert (int(r["nac"]), int(r["duid"])) == (k >> 4, k & 15)
        if nerr <= 11:
            assert k == data and dist == nerr


def test_predecimator_against_the_batch_model(spec):
    """SPEC 3.0 (the 2.4 Msps -> 240 ksps stage of config 3): the oracle's streaming object in ragged chunks against the
    whole-array restatement, bit for bit."""
    import spec_model
    rng = np.random.default_rng(4)
    x = (rng.normal(0, 0.3, 50003) + 1j * rng.normal(0, 0.3, 50003)).astype(np.complex64)
    p = O.PreDecim()
    got = np.concatenate([p.feed(x[o:o + 7777]) for o in range(0, len(x), 7777)])
    h0 = spec["pre_taps"]
    ref_r = spec_model._fir(x.real.astype(np.float32), h0, int(spec["pre_decim"]), int(spec["pre_decim"]) - 1)
    ref_i = spec_model._fir(x.imag.astype(np.float32), h0, int(spec["pre_decim"]), int(spec["pre_decim"]) - 1)
    assert len(got) == len(ref_r) == 5000
    assert np.array_equal(got.real.view(np.uint32), ref_r.view(np.uint32)) and np.array_equal(got.imag.view(np.uint32), ref_i.view(np.uint32))


def test_tracking_clock_range_of_usefulness():
    """docs/SPEC.md 3.8b's table, two of its rows: at 150 ppm with P25's 0.18 s between sync words the tracking clock makes a
    handful of symbol errors where the fixed stride makes hundreds; at 20 ppm re-anchoring alone is error free, and so is
    the tracking clock now that its period comes from sync positions refined to quarter samples (the whole-sample period
    made 20 errors here)."""
    def errors(ppm, mode):
        iq, truth, _ = c4fm.synth(4.0, seed=5, snr_db=30.0, clock_ppm=ppm)
        d = O.Recv(O.make_config(symbol_clock=mode)).feed(O.Demod().feed_cf32(iq))[0]
        k = min(len(d), len(truth) - 24)
        return int(np.count_nonzero(d[:k] != truth[24:24 + k])), k
    f150, k = errors(150.0, 0)
    t150, _ = errors(150.0, 1)
    f20, _ = errors(20.0, 0)
    t20, _ = errors(20.0, 1)
    assert k > 19000 and f150 > 50 and t150 < f150 // 8
    assert f20 == 0 and t20 == 0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_densest_detections_against_the_independent_batch_model(spec, mode):
    """The ten polyphase planes of the baseband are independent sample sets: every plane carrying its own back-to-back sync words, plane
    r's ending at positions 240 m + 21 r, makes ten detections per 240 samples -- 21 or 51 samples apart, 320 per 7 680 samples, denser
    than any transmitter can be and close to SPEC 3.7's bound (detections are more than 5 samples apart).  The oracle's receivers (the
    GPU tests' reference for this scene, tests/test_gpu_recv_general.py) against the batch model, all three clocks, with lock drops."""
    import spec_model
    mask = int(spec["sync_sign_mask"]) if "sync_sign_mask" in spec else 0x050cdf
    pat = np.array([1.0 if (mask >> j) & 1 else -1.0 for j in range(24)], dtype=np.float32)
    n_bb = 2 * 7680 + 333
    q = np.arange(n_bb, dtype=np.int64)
    bb = (0.24 * pat[((q - 21 * (q % 10) + 230) // 10) % 24]).astype(np.float32)
    bb += (0.004 * np.random.default_rng(11).standard_normal(n_bb)).astype(np.float32)
    for drops in ([], [1000, 1007, 5000, 7685, 12001]):
        if mode == 2:
            got = O.recv_range(bb, O.make_config(symbol_clock=2), drops)
            ref = spec_model.Model(spec).receive_tracking(bb, spec, drops, reslice=True)
        else:
            r = O.Recv(O.make_config(symbol_clock=mode))
            outs, o = [], 0
            for c in sorted(set(drops + [n_bb])):
                outs.append(r.feed(bb[o:c]))
                if c in drops:
                    r.resync()
                o = c
            got = [np.concatenate([x[k] for x in outs]) for k in range(3)]
            m = spec_model.Model(spec)
            ref = m.receive_tracking(bb, spec, drops) if mode == 1 else m.receive(bb, drops)
        assert len(got[1]) > 600
        for k in range(3):
            assert len(got[k]) == len(ref[k]) and np.array_equal(got[k], np.asarray(ref[k]).astype(got[k].dtype)), (mode, drops, k)
