"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle, bit-exact.

Bar (BASELINE.json north_star): dibits, sync positions and -- because both sides implement the
same fp32 operation sequence (docs/SPEC.md section 3) -- the baseband floats are compared bit for
bit.  Only power_dbm (tree vs sequential reduction) has a tolerance: 1e-3 dB.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def FE():
    from p25rx_amd.frontend import FrontEnd
    return FrontEnd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_library_loaded_is_in_tree():
    from p25rx_amd import _lib
    L = _lib.load()
    assert os.path.samefile(L._name, os.path.join(os.path.dirname(_lib.__file__), "libp25fe.so"))


def test_demod_cf32_reference_chunks_bit_exact(O, FE, c4fm_1s):
    """DemodTask::run in the reference's 16384-sample chunks (src/consts.rs:8): lengths 3276/3277 and bits."""
    iq, _, _ = c4fm_1s
    fe, od = FE(), O.Demod()
    for off in range(0, len(iq), 16384):
        a = fe.demod_cf32(iq[off:off + 16384])
        b = od.feed_cf32(iq[off:off + 16384])
        assert len(a) == len(b)
        assert np.array_equal(bits(a), bits(b)), "chunk at %d" % off


def test_demod_u8_golden_fixture(O, FE):
    g = np.load(os.path.join(GOLDEN, "c4fm_seed7_u8.npz"))
    fe = FE()
    bb = np.concatenate([fe.demod_u8(g["iq_u8"][o:o + 32768]) for o in range(0, len(g["iq_u8"]), 32768)])
    assert np.array_equal(bits(bb), g["bb_bits"])
    dib, spos, sdib = fe.slice(bb)
    assert np.array_equal(dib, g["dibits"])
    assert np.array_equal(spos, g["sync_pos"])
    assert np.array_equal(sdib, g["sync_dibit"])


@pytest.mark.parametrize("seed", [3, 4])
def test_demod_ragged_chunks(O, FE, c4fm_1s, seed):
    """Any chunking gives the same stream (state carry of src/demod.rs:25-40), including 1-sample and empty chunks."""
    iq, _, _ = c4fm_1s
    iq = iq[:100000]
    ref = O.Demod().feed_cf32(iq)
    fe = FE()
    rng = np.random.default_rng(seed)
    parts, off = [], 0
    sizes = [0, 1, 1, 2, 3, 4, 5, 7, 283, 284, 285, 1279, 1280, 1281]
    while off < len(iq):
        n = sizes.pop(0) if sizes else int(rng.integers(1, 20000))
        parts.append(fe.demod_cf32(iq[off:off + n]))
        off += n
    assert np.array_equal(bits(np.concatenate(parts)), bits(ref))


def test_demod_u8_ragged(O, FE, c4fm_1s):
    from p25rx_amd import c4fm
    u8 = c4fm.to_u8(c4fm_1s[0][:60000])
    ref = O.Demod().feed_u8(u8)
    fe = FE()
    rng = np.random.default_rng(8)
    parts, off = [], 0
    while off < len(u8):
        n = 2 * int(rng.integers(1, 9000))
        parts.append(fe.demod_u8(u8[off:off + n]))
        off += n
    assert np.array_equal(bits(np.concatenate(parts)), bits(ref))


def test_kat_tone_and_dc(FE):
    n = 48000
    t = np.arange(n) / 240000.0
    bb = FE().demod_cf32((0.5 * np.ones(n)).astype(np.complex64))
    assert np.all(bb[80:] == 0.0)
    bb = FE().demod_cf32((0.5 * np.exp(2j * np.pi * 1800.0 * t)).astype(np.complex64))
    assert np.abs(bb[80:] - 0.36).max() < 5e-6


def test_power_dbm(O, FE, c4fm_1s):
    """power_dbm (src/demod.rs:123-134): tree reduction on the GPU vs the reference's sequential fold, 1e-3 dB."""
    iq = c4fm_1s[0][:16384 * 3]
    fe, od = FE(), O.Demod()
    for off in range(0, len(iq), 16384):
        _, pa = fe.demod_cf32(iq[off:off + 16384], want_power=True)
        _, pb = od.feed_cf32(iq[off:off + 16384], want_power=True)
        assert abs(pa - pb) < 1e-3
    x = (0.25 * np.exp(1j * np.linspace(0, 900, 16384))).astype(np.complex64)
    _, p = FE().demod_cf32(x, want_power=True)
    _, po = O.Demod().feed_cf32(x, want_power=True)
    assert abs(p - po) < 1e-3
    assert abs(p - (30 + 20 * np.log10(0.25))) < 0.3      # 2.1 kHz tone: passband droop + start-up transient


def test_slice_ragged_chunks_and_sync_events(O, FE, c4fm_1s):
    iq = c4fm_1s[0]
    bb = O.Demod().feed_cf32(iq)
    ref = O.Recv().feed(bb)
    fe = FE()
    rng = np.random.default_rng(5)
    outs, off = [], 0
    sizes = [1, 1, 5, 6, 10, 229, 230, 231, 241, 2047, 2048, 2049]
    while off < len(bb):
        n = sizes.pop(0) if sizes else int(rng.integers(1, 9000))
        outs.append(fe.slice(bb[off:off + n]))
        off += n
    assert np.array_equal(np.concatenate([o[0] for o in outs]), ref[0])
    assert np.array_equal(np.concatenate([o[1] for o in outs]), ref[1])
    assert np.array_equal(np.concatenate([o[2] for o in outs]), ref[2])


def test_resync_matches_oracle(O, FE, c4fm_1s):
    """MessageReceiver::resync (src/recv.rs:136,179) at several stream positions, including right at a sync word."""
    bb = O.Demod().feed_cf32(c4fm_1s[0])
    for k in (20000, 8942 + 6, 8942 + 5, 8942 + 4, 8942, 300, 17582 + 11):
        fe, orc = FE(), O.Recv()
        a1, b1 = fe.slice(bb[:k]), orc.feed(bb[:k])
        fe.resync()
        orc.resync()
        a2, b2 = fe.slice(bb[k:]), orc.feed(bb[k:])
        for x, y in zip(a1 + a2, b1 + b2):
            assert np.array_equal(x, y), "resync at %d" % k


def test_run_fused_host_matches_oracle(O, FE, c4fm_1s):
    iq, truth, _ = c4fm_1s
    ref = O.run_cf32(iq)
    fe = FE()
    got = np.concatenate([fe.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
    assert np.array_equal(got, ref)
    n = min(len(got), len(truth) - 24)
    assert np.array_equal(got[:n], truth[24:24 + n])      # and both equal the modulator's symbols


def test_run_dev_matches_oracle_10s(O, FE):
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, truth, _ = c4fm.synth(10.0, seed=21, snr_db=20.0)
    ref = O.run_cf32(iq)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    dib, res = fe.run_dev(t)
    r = parse_results(res)[0]
    got = dib[0, :int(r["n_dibits"])].cpu().numpy()
    assert int(r["n_baseband"]) == len(iq) // 5
    assert np.array_equal(got, ref)
    # second call on the same handle: fresh-stream semantics, identical output
    dib2, res2 = fe.run_dev(t)
    assert torch.equal(dib2[0, :len(got)], dib[0, :len(got)])


def test_demod_dev_u8_and_cf32_vs_oracle(O, FE, c4fm_1s):
    import torch
    from p25rx_amd import c4fm
    iq = c4fm_1s[0][:123457]
    ref = O.Demod().feed_cf32(iq)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    bb, nb = FE().demod_dev(t)
    assert nb == len(ref)
    assert np.array_equal(bits(bb[0, :nb].cpu().numpy()), bits(ref))
    u8 = c4fm.to_u8(iq)
    ref8 = O.Demod().feed_u8(u8)
    t8 = torch.from_numpy(u8.reshape(-1, 2)).cuda()
    bb8, nb8 = FE().demod_dev(t8)
    assert np.array_equal(bits(bb8[0, :nb8].cpu().numpy()), bits(ref8))


def test_multichannel_channel_major(O, FE):
    """BASELINE.json config 4 layout: [C][n] channel-major, independent per-channel state."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    Cn = 5
    iqs = [c4fm.synth(0.4, seed=100 + c, snr_db=18.0, timing_offset=7 * c, freq_offset_hz=50.0 * c)[0] for c in range(Cn)]
    arr = np.stack(iqs)
    refs = [O.run_cf32(x) for x in iqs]
    t = torch.from_numpy(arr.view(np.float32).reshape(Cn, -1, 2)).cuda()
    fe = FE(n_channels=Cn)
    dib, res = fe.run_dev(t)
    r = parse_results(res)
    for c in range(Cn):
        assert np.array_equal(dib[c, :int(r["n_dibits"][c])].cpu().numpy(), refs[c])
    # streaming host path, multi-channel
    fe2 = FE(n_channels=Cn)
    parts = [fe2.run_cf32(arr[:, o:o + 30000]) for o in range(0, arr.shape[1], 30000)]
    for c in range(Cn):
        assert np.array_equal(np.concatenate([p[c] for p in parts]), refs[c])


def test_state_export_import(O, FE, c4fm_1s):
    iq = c4fm_1s[0]
    ref = O.run_cf32(iq)
    fe = FE()
    a = fe.run_cf32(iq[:100001])
    blob = fe.state_export()
    fe2 = FE()
    fe2.state_import(blob)
    b = fe2.run_cf32(iq[100001:])
    assert np.array_equal(np.concatenate([a, b]), ref)
    # a blob is outside data: a damaged one is refused and leaves the handle as it was
    from p25rx_amd._lib import P25feError
    fe3 = FE()
    c = fe3.run_cf32(iq[:100001])
    for off, val in ((0, 0), (12, 9)):                               # magic; fmt_locked
        bad = np.array(blob, dtype=np.uint8, copy=True)
        bad[off] = val
        with pytest.raises(P25feError):
            fe3.state_import(bad)
    with pytest.raises(P25feError):
        fe3.state_import(blob[:-1])
    d = fe3.run_cf32(iq[100001:])
    assert np.array_equal(np.concatenate([c, d]), ref)


def test_time_shards_equal_single_pass(O, FE):
    """BASELINE.json config 5: contiguous time shards with left halo + anchor hand-off == one pass."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results, n_baseband
    iq, _, _ = c4fm.synth(2.0, seed=33, snr_db=20.0, frame_dibits=1500)
    ref = O.run_cf32(iq)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    halo = fe.shard_halo()
    cuts = [0, 100004, 100004 + 3002, 310006, len(iq)]   # uneven, even cut points (16-B loads), one short shard
    fes = [FE() for _ in range(len(cuts) - 1)]
    summ, bb0, bbn = [], [], []
    for r in range(len(cuts) - 1):
        a, b = cuts[r], cuts[r + 1]
        h = min(a, halo)
        if r % 2:                                               # the two-launch form (halo exchange hidden behind K1)
            fes[r].shard_pass1_main(t[a - h:b], offset=h, n_hist=h, abs0=a)
            if r == 3:                                          # ... and the three-launch form: the head on its own (another stream's)
                fes[r].shard_pass1_head(t[a - h:b], offset=h, n_hist=h, abs0=a)
            res = fes[r].shard_pass1_finish(t[a - h:b], offset=h, n_hist=h, abs0=a)
        else:
            res = fes[r].shard_pass1(t[a - h:b], offset=h, n_hist=h, abs0=a)
        summ.append(parse_results(res)[0])
        bb0.append(n_baseband(0, a))
        bbn.append(n_baseband(a, b - a))
    anc, off = fe.shard_resolve(np.array(summ), bb0, bbn)
    # pass 2 with the combine INSIDE the slicer (p25fe_shard_pass2_dev: no second scan, no resolve launch), straight after
    # pass 1 as the product runs it: rank 0 also slices into the ordered stream, the rest is compacted behind it
    summ_t = torch.from_numpy(np.frombuffer(np.array(summ).tobytes(), dtype=np.uint8).copy()).view(len(summ), -1).cuda()
    d_bb0 = torch.tensor(bb0, dtype=torch.int64, device="cuda")
    d_bbn = torch.tensor(bbn, dtype=torch.int64, device="cuda")
    cap = max((b // 10 + 64 + 15) // 16 * 16 for b in bbn)
    gathered = torch.zeros((len(cuts) - 1, cap), dtype=torch.uint8, device="cuda")
    stream = torch.full((len(ref) + 64,), 255, dtype=torch.uint8, device="cuda")
    fused = []
    for r in range(len(cuts) - 1):
        dib, res, anc3, off3 = fes[r].shard_pass2_dev(summ_t, d_bb0, d_bbn, r, bbn[r], dibits=gathered[r:r + 1],
                                                      dup=(stream[None, :] if r == 0 else None))
        assert np.array_equal(off3.cpu().numpy().astype(np.uint64), off) and anc3.cpu().numpy().tobytes() == anc.tobytes()
        fused.append((parse_results(res)[0], gathered[r, :int(parse_results(res)[0]["n_dibits"])].cpu().numpy()))
    assert np.array_equal(np.concatenate([f[1] for f in fused]), ref)
    fe.shard_compact_from_dev(gathered, torch.from_numpy(off.astype(np.int64)).cuda(), 1, stream)
    assert np.array_equal(stream[:len(ref)].cpu().numpy(), ref) and bool((stream[len(ref):] == 255).all())
    out = []
    for r in range(len(cuts) - 1):
        dib, res = fes[r].shard_pass2(anc[r:r + 1], bbn[r], t.device)
        assert parse_results(res)[0].tobytes() == fused[r][0].tobytes(), (r, parse_results(res)[0], fused[r][0])   # the same final record
        k = int(parse_results(res)[0]["n_dibits"])
        assert off[r] == sum(len(x) for x in out)
        out.append(dib[0, :k].cpu().numpy())
    assert np.array_equal(np.concatenate(out), ref)
    # the same combine on the device (no host sync), and pass 2 fed with the device anchors
    # (pass 2 above overwrote nothing the summaries depend on: they were parsed before)
    d_anc, d_off = fe.shard_resolve_dev(summ_t, d_bb0, d_bbn)
    assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), off)
    assert d_anc.cpu().numpy().tobytes() == anc.tobytes()
    out2 = []
    for r in range(len(cuts) - 1):
        dib, res = fes[r].shard_pass2(d_anc[r:r + 1], bbn[r], t.device)
        out2.append(dib[0, :int(parse_results(res)[0]["n_dibits"])].cpu().numpy())
    assert np.array_equal(np.concatenate(out2), ref)
    # ... and p25fe_shard_pass2_dev AFTER a p25fe_shard_pass2 (which rewrote pass 1's scan): falls back to resolve + scan + slicer
    for r in range(len(cuts) - 1):
        dib, res, _, _ = fes[r].shard_pass2_dev(summ_t, d_bb0, d_bbn, r, bbn[r])
        assert parse_results(res)[0].tobytes() == fused[r][0].tobytes()
        assert np.array_equal(dib[0, :len(out2[r])].cpu().numpy(), out2[r])


def test_time_shard_steps_pipelined(O, FE):
    """p25fe_shard_pipe_begin / _end (what p25fe_shard_step_pipelined drives): steps over ALTERNATING captures enqueued back to back
    -- K1's main launch on the caller's stream, the head on a side stream (the detection's first tiles poll its flag), detection /
    scan / combine + slicer on the handle's receive stream two scratch sets deep -- give, step for step, the bytes of the same
    shard procedure run one call at a time."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results, n_baseband
    caps, refs = [], []
    for seed, fd in ((41, 1500), (42, 700)):
        iq, _, _ = c4fm.synth(2.0, seed=seed, snr_db=20.0, frame_dibits=fd)
        caps.append(torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda())
        refs.append(O.run_cf32(iq))
    n_all = caps[0].shape[0]
    cut = 200008
    halo = FE().shard_halo()
    bb0 = [n_baseband(0, 0), n_baseband(0, cut)]
    bbn = [n_baseband(0, cut), n_baseband(cut, n_all - cut)]
    d_bb0 = torch.tensor(bb0, dtype=torch.int64, device="cuda")
    d_bbn = torch.tensor(bbn, dtype=torch.int64, device="cuda")
    # shard 0's summary per capture (a plain pass 1 on its own handle); shard 1 is the pipelined one
    fe0, summ0 = FE(), []
    for t in caps:
        summ0.append(fe0.shard_pass1(t[:cut], offset=0, n_hist=0, abs0=0).clone())
    torch.cuda.synchronize()

    def plain(fe, t):
        res = fe.shard_pass1(t[cut - halo:], offset=halo, n_hist=halo, abs0=cut)
        return res

    # the one-call-at-a-time answer of shard 1 for both captures
    want = []
    fe1 = FE()
    for w, t in enumerate(caps):
        res = plain(fe1, t)
        summ_t = torch.cat([summ0[w], res])
        dib, res2, _, off = fe1.shard_pass2_dev(summ_t, d_bb0, d_bbn, 1, bbn[1])
        r = parse_results(res2)[0]
        off = off.cpu().numpy()
        assert int(off[2]) == len(refs[w])
        want.append((r.tobytes(), dib[0, :int(r["n_dibits"])].cpu().numpy().copy()))
        assert np.array_equal(want[-1][1], refs[w][int(off[1]):])
    # pipelined: every step enqueued before the first one has finished
    fe = FE()
    side = torch.cuda.Stream()
    st = torch.cuda.current_stream()
    order = [0, 1, 1, 0, 1, 0, 0, 1]
    cap = (bbn[1] // 10 + 64 + 15) // 16 * 16
    outs = [(torch.zeros((1, cap), dtype=torch.uint8, device="cuda"), torch.zeros_like(summ0[0])) for _ in order]
    for k, w in enumerate(order):
        t = caps[w][cut - halo:]
        rx = fe.shard_pipe_begin()
        fork = st.record_event()                                  # the side stream may not overtake the wait _begin put on `st`
        fe.shard_pass1_main(t, offset=halo, n_hist=halo, abs0=cut)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            fe.shard_pass1_head(t, offset=halo, n_hist=halo, abs0=cut)
        with torch.cuda.stream(rx):
            res = fe.shard_pass1_finish(t, offset=halo, n_hist=halo, abs0=cut)
            summ_t = torch.cat([summ0[w], res])
            fe.shard_pass2_dev(summ_t, d_bb0, d_bbn, 1, bbn[1], dibits=outs[k][0], result=outs[k][1])
        fe.shard_pipe_end()
    fe.join_dev()
    torch.cuda.synchronize()
    for k, w in enumerate(order):
        r = parse_results(outs[k][1])[0]
        assert r.tobytes() == want[w][0], (k, w)
        assert np.array_equal(outs[k][0][0, :int(r["n_dibits"])].cpu().numpy(), want[w][1]), (k, w)
    # the same steps with the WHOLE front end in one launch on the caller's stream (p25fe_shard_pass1_k1: what the pipelined RCCL step
    # runs once the halo has arrived in front of K1), detection + scan on the receive stream, pass 2 on a third stream behind an event
    outs2 = [(torch.zeros((1, cap), dtype=torch.uint8, device="cuda"), torch.zeros_like(summ0[0])) for _ in order]
    for k, w in enumerate(order):
        t = caps[w][cut - halo:]
        rx = fe.shard_pipe_begin()
        fe.shard_pass1_k1(t, offset=halo, n_hist=halo, abs0=cut)
        with torch.cuda.stream(rx):
            res = fe.shard_pass1_finish(t, offset=halo, n_hist=halo, abs0=cut)
            summ_t = torch.cat([summ0[w], res])
            stage = rx.record_event()
        with torch.cuda.stream(side):
            side.wait_event(stage)
            fe.shard_pass2_dev(summ_t, d_bb0, d_bbn, 1, bbn[1], dibits=outs2[k][0], result=outs2[k][1])
            summ_t.record_stream(side)
        fe.shard_pipe_end(last=side)
    fe.join_dev()
    torch.cuda.synchronize()
    for k, w in enumerate(order):
        r = parse_results(outs2[k][1])[0]
        assert r.tobytes() == want[w][0], ("k1", k, w)
        assert np.array_equal(outs2[k][0][0, :int(r["n_dibits"])].cpu().numpy(), want[w][1]), ("k1", k, w)
    # a plain call right behind a pipelined step joins the receive stream by itself
    rx = fe.shard_pipe_begin()
    fe.shard_pass1_main(caps[1][cut - halo:], offset=halo, n_hist=halo, abs0=cut)
    with torch.cuda.stream(rx):
        res = fe.shard_pass1_finish(caps[1][cut - halo:], offset=halo, n_hist=halo, abs0=cut)
        summ_t = torch.cat([summ0[1], res])
        fe.shard_pass2_dev(summ_t, d_bb0, d_bbn, 1, bbn[1], dibits=outs[0][0], result=outs[0][1])
    fe.shard_pipe_end()
    dib, res = fe.run_dev(caps[0])
    torch.cuda.synchronize()
    assert np.array_equal(dib[0, :len(refs[0])].cpu().numpy(), refs[0])
    assert parse_results(outs[0][1])[0].tobytes() == want[1][0]
    # protocol errors: a second _begin, an _end without _begin
    from p25rx_amd._lib import P25feError
    fe.shard_pipe_begin()
    with pytest.raises(P25feError):
        fe.shard_pipe_begin()
    fe.shard_pipe_end()
    with pytest.raises(P25feError):
        fe.shard_pipe_end()


def test_custom_taps_zero_padded(O, FE, c4fm_1s):
    """Shorter filters are accepted (zero-padded at the old end) and still match the oracle bit for bit."""
    spec = O.load_spec()
    dt = spec["decim_taps"][:21]
    ct = spec["chan_taps"][:33]
    iq = c4fm_1s[0][:50000]
    ref = O.Demod(O.make_config(spec, dt, ct)).feed_cf32(iq)
    got = FE(decim_taps=dt, chan_taps=ct).demod_cf32(iq)
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("nt", [(64, 64), (47, 53), (32, 41), (31, 42)])
def test_long_taps_up_to_the_abi_ceiling(O, FE, c4fm_1s, nt):
    """p25fe_config_t carries up to 64 taps per filter so that the reference's own p25_filts tables (src/demod.rs:27-29,
    sizes unknown here) can be loaded: random, asymmetric tables of 64 + 64 (and odd sizes, and one tap past the build's
    own 31 / 41) match the oracle bit for bit -- streaming host calls in ragged chunks, u8 and cf32, device ranges with a
    history, and the fused path's dibits."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    rng = np.random.default_rng(100 + nt[0])
    dt = (rng.standard_normal(nt[0]) * np.hanning(nt[0] + 2)[1:-1] / 6).astype(np.float32).tolist()
    ct = (rng.standard_normal(nt[1]) * np.hanning(nt[1] + 2)[1:-1] / 6).astype(np.float32).tolist()
    cfg = O.make_config(O.load_spec(), dt, ct)
    iq = c4fm_1s[0]
    ref = O.Demod(cfg).feed_cf32(iq)
    fe = FE(decim_taps=dt, chan_taps=ct)
    cuts = [0, 1, 17, 3276, 16384, 16385, 50001, 123456, len(iq)]
    got = np.concatenate([fe.demod_cf32(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(bits(got), bits(ref))
    u8 = c4fm.to_u8(iq)
    ref8 = O.Demod(cfg).feed_u8(u8)
    fe8 = FE(decim_taps=dt, chan_taps=ct)
    got8 = np.concatenate([fe8.demod_u8(u8[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(bits(got8), bits(ref8))
    # device range that starts mid-stream with the history in memory
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    a0 = 100000
    h = 448
    bb, nb = FE(decim_taps=dt, chan_taps=ct).demod_dev(t[a0 - h:], n_hist=h, abs0=a0, offset=h)
    assert np.array_equal(bits(bb[0, :nb].cpu().numpy()), bits(ref[len(ref) - nb:]))
    # fused path (planar K1 of the long-tap geometry) vs the oracle's receiver on the oracle's baseband
    dib_ref = O.Recv().feed(ref)[0]
    dib, res = FE(decim_taps=dt, chan_taps=ct).run_dev(t)
    n = int(parse_results(res)[0]["n_dibits"])
    assert n == len(dib_ref) and np.array_equal(dib[0, :n].cpu().numpy(), dib_ref)


def test_errors_are_loud(FE):
    from p25rx_amd._lib import ANCHOR_DTYPE, P25feError
    fe = FE()
    fe.demod_cf32(np.zeros(100, dtype=np.complex64))
    with pytest.raises(P25feError):
        fe.demod_u8(np.zeros(100, dtype=np.uint8))        # format switch inside a stream
    with pytest.raises(P25feError):
        FE(device=99)
    with pytest.raises(P25feError):
        FE(decim_taps=[0.1] * 65)                              # more taps than the ABI carries
    # sharded passes out of order fail instead of working on stale scratch
    import torch
    t = torch.zeros((48000, 2), dtype=torch.float32, device="cuda")
    fe2 = FE()
    with pytest.raises(P25feError):
        fe2.shard_pass1_finish(t, offset=0, n_hist=0, abs0=0)   # no matching shard_pass1_main
    with pytest.raises(P25feError):
        fe2.shard_pass2(np.zeros(1, dtype=ANCHOR_DTYPE), 9600, t.device)   # no pass 1 at all
    fe2.shard_pass1_main(t, offset=0, n_hist=0, abs0=0)
    with pytest.raises(P25feError):
        fe2.shard_pass1_finish(t[:24000], offset=0, n_hist=0, abs0=0)   # a different shard than the main launch


def test_full_size_property_60s(FE):
    """Size-independent property at a BASELINE-scale input: demod(mod(d)) == d on 60 s generated in HBM."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    n = 60 * 240000
    iq, truth = c4fm.synth_torch(n, seed=5, device="cuda", snr_db=30.0)
    dib, res = FE().run_dev(iq)
    r = parse_results(res)[0]
    got = dib[0, :int(r["n_dibits"])].cpu().numpy()
    assert int(r["n_sync"]) == len(truth) // 864 + (1 if len(truth) % 864 >= 24 else 0)
    k = min(len(got), len(truth) - 24)
    assert k > 287000
    assert np.array_equal(got[:k], truth[24:24 + k])


def test_config3_wideband_predecimator(O, FE):
    """BASELINE.json config 3: 2.4 Msps capture -> 10:1 decimating FIR -> the 240 ksps chain, bit-exact vs the oracle.
    The wideband signal is the C4FM channel interpolated x10 plus a strong interferer 300 kHz away."""
    import torch
    from scipy import signal as sps
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, truth, _ = c4fm.synth(0.5, seed=12, snr_db=25.0)
    wide = sps.resample_poly(iq.astype(np.complex128), 10, 1)
    t = np.arange(len(wide)) / 2.4e6
    wide = (wide + 0.8 * np.exp(2j * np.pi * 300e3 * t)).astype(np.complex64)
    x240 = O.PreDecim().feed(wide)
    ref = O.run_cf32(x240)
    k = min(len(ref), len(truth) - 24)
    assert np.array_equal(ref[:k], truth[24:24 + k])                 # the interferer is gone, the symbols survive
    tw = torch.from_numpy(wide.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    y, no = fe.predecim_dev(tw)
    assert no == len(x240)
    assert np.array_equal(y[0, :no].cpu().numpy().view(np.uint32), x240.view(np.float32).reshape(-1, 2).view(np.uint32))
    dib, res = fe.run_dev(y[:, :no])
    assert np.array_equal(dib[0, :int(parse_results(res)[0]["n_dibits"])].cpu().numpy(), ref)
    # a range in the middle of the capture with history, decimation grid not aligned to the range start
    off, n_hist = 100008, 96
    y2, no2 = fe.predecim_dev(tw, n_hist=n_hist, abs0=off, offset=off)
    first = len([m for m in range(len(x240)) if 10 * m + 9 < off])   # outputs produced before the range
    assert np.array_equal(y2[0, :no2].cpu().numpy().view(np.uint32),
                          x240[first:first + no2].view(np.float32).reshape(-1, 2).view(np.uint32))


def test_config4_256_channels(O, FE):
    """BASELINE.json config 4 shape: 256 independent channels, channel-major, one launch per kernel; every
    channel must equal its own oracle run (distinct seeds, SNRs, frequency and timing offsets)."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    Cn, secs = 256, 0.2
    iqs = []
    for c in range(Cn):
        iq, _, _ = c4fm.synth(secs, seed=500 + c, snr_db=12.0 + (c % 20), freq_offset_hz=20.0 * (c % 11) - 100.0,
                              timing_offset=c % 50, frame_dibits=300 + 7 * (c % 13))
        iqs.append(iq)
    arr = np.stack(iqs)
    t = torch.from_numpy(arr.view(np.float32).reshape(Cn, -1, 2)).cuda()
    fe = FE(n_channels=Cn)
    dib, res = fe.run_dev(t)
    r = parse_results(res)
    for c in range(Cn):
        ref = O.run_cf32(iqs[c])
        assert int(r["n_dibits"][c]) == len(ref), c
        assert np.array_equal(dib[c, :len(ref)].cpu().numpy(), ref), c


def test_fuzz_ranges_and_alignment(O, FE, c4fm_1s):
    """Randomised device ranges: any (offset, n_hist, length) on the 16-B grid reproduces the oracle stream exactly,
    for cf32 and u8; unaligned pointers and bad strides are rejected with P25FE_ERR_ARG, never silently mis-read."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd._lib import P25feError, ERR_ARG
    iq = c4fm_1s[0][:150000]
    ref = O.Demod().feed_cf32(iq)
    u8 = c4fm.to_u8(iq)
    ref8 = O.Demod().feed_u8(u8)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    t8 = torch.from_numpy(u8.reshape(-1, 2)).cuda()
    rng = np.random.default_rng(42)
    fe = FE()
    for trial in range(24):
        use8 = trial % 2 == 1
        g = 8 if use8 else 2
        off = int(rng.integers(150, 60000)) // g * g
        n = int(rng.integers(1, 80000))
        n = min(n, len(iq) - off)
        n_hist = int(rng.choice([288, 300, 1000, off]))
        n_hist = min(n_hist, off)
        src, rr = (t8, ref8) if use8 else (t, ref)
        bb, nb = fe.demod_dev(src[:off + n], n_hist=n_hist, abs0=off, offset=off)
        first = len([m for m in range(off // 5 - 1, off // 5 + 2) if 5 * m + 4 < off]) + max(off // 5 - 1, 0)
        got = bb[0, :nb].cpu().numpy()
        # outputs produced by inputs inside [off, off + n): exact provided the history covers the filter memory
        assert np.array_equal(bits(got), bits(rr[first:first + nb])), (trial, off, n, n_hist)
    with pytest.raises(P25feError) as e:
        fe.demod_dev(t[1:], n_hist=0, abs0=0)                      # owned sample 0 at an 8-byte, not 16-byte, address
    assert e.value.status == ERR_ARG
    fe3 = FE(n_channels=3)
    with pytest.raises(P25feError):
        bad = torch.zeros((3, 1001, 2), dtype=torch.float32, device="cuda")   # odd channel stride
        fe3.demod_dev(bad)


def test_range_longer_than_the_kernels_index_is_an_argument_error(FE):
    """A range of 2^31 symbols or more (124 h of one channel) is refused as an argument error before anything is sized
    for it -- the sample pointer is never followed."""
    import ctypes as C
    import torch
    from p25rx_amd._lib import ERR_ARG, RESULT_DTYPE
    fe = FE()
    x = torch.zeros((64, 2), dtype=torch.float32, device="cuda")
    dib = torch.zeros((1, 64), dtype=torch.uint8, device="cuda")
    res = torch.zeros((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    n = 5 * (0x7ff00000 * 10 + 5)
    args = (fe.h, C.c_void_p(x.data_ptr()), 0, 0, n, C.c_void_p(dib.data_ptr()), 64, C.c_void_p(res.data_ptr()), fe._stream())
    assert fe.L.p25fe_run_dev(*args) == ERR_ARG
    assert fe.L.p25fe_run_dev_pipelined(*args) == ERR_ARG
    assert fe.L.p25fe_shard_pass1(fe.h, C.c_void_p(x.data_ptr()), 0, 0, 0, n, 0, C.c_void_p(res.data_ptr()), fe._stream()) == ERR_ARG
    assert fe.L.p25fe_slice_dev(fe.h, C.c_void_p(x.data_ptr()), 0, 0, n // 5, 0, None, C.c_void_p(dib.data_ptr()), 64, None, None, 0,
                                C.c_void_p(res.data_ptr()), fe._stream()) == ERR_ARG
    # the handle is as good as before
    d2, r2 = fe.run_dev(x)
    torch.cuda.synchronize()


def test_tiny_and_empty_inputs(O, FE):
    """Empty, 1-sample and sub-filter-length inputs through every host entry point."""
    fe = FE()
    assert len(fe.demod_cf32(np.zeros(0, np.complex64))) == 0
    assert len(fe.run_cf32(np.zeros(0, np.complex64))) == 0
    d, sp, sd = fe.slice(np.zeros(0, np.float32))
    assert len(d) == 0 and len(sp) == 0
    x = (np.arange(1, 41) * (0.01 + 0.02j)).astype(np.complex64)
    od, fe2 = O.Demod(), FE()
    for k in range(0, 40, 3):
        assert np.array_equal(bits(fe2.demod_cf32(x[k:k + 3])), bits(od.feed_cf32(x[k:k + 3])))
    orc, fe3 = O.Recv(), FE()
    b = np.linspace(-0.4, 0.4, 77).astype(np.float32)
    for k in range(77):
        a1, a2 = fe3.slice(b[k:k + 1]), orc.feed(b[k:k + 1])
        assert all(np.array_equal(p, q) for p, q in zip(a1, a2))


def test_non_finite_and_extreme_samples(O, FE, c4fm_1s):
    """A damaged capture: NaN, +-Inf, 1e30, denormals and -0.0 sprinkled into the IQ stream.  Every baseband sample that is a
    number must equal the oracle's bit for bit, every NaN must be a NaN on both sides (the payload and sign of a generated
    NaN are the hardware's), and the receiver's output on the parts of the stream the damage does not reach must agree."""
    import torch
    iq = np.array(c4fm_1s[0][:120000], copy=True)
    f = iq.view(np.float32)
    rng = np.random.default_rng(5)
    vals = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-42, -1e-42, -0.0, 3.0e38], dtype=np.float32)
    hits = np.sort(rng.choice(np.arange(20000, 60000), size=40, replace=False))
    for k, i in enumerate(hits):
        f[2 * i + (k & 1)] = vals[k % len(vals)]
    ref = O.Demod().feed_cf32(iq)
    fe = FE()
    got = fe.demod_cf32(iq)
    assert len(got) == len(ref)
    nan_r, nan_g = np.isnan(ref), np.isnan(got)
    assert np.array_equal(nan_r, nan_g) and nan_r.any()
    assert np.array_equal(bits(got[~nan_g]), bits(ref[~nan_r]))
    # the device-resident form (polyphase planes) and the streaming form see the same floats
    t = torch.from_numpy(f.reshape(-1, 2)).cuda()
    bb, nb = fe.demod_dev(t)
    g2 = bb[0, :nb].cpu().numpy()
    assert np.array_equal(np.isnan(g2), nan_r) and np.array_equal(bits(g2[~nan_r]), bits(ref[~nan_r]))
    # the receiver on it: a NaN fails every comparison on both sides alike, so the streams agree THROUGH the damage, under
    # both clocks, from the oracle's baseband, from the GPU's (its NaNs carry the other sign bit) and fused
    for mode in (0, 1):
        ro = O.Recv(O.make_config(symbol_clock=mode)).feed(ref)
        assert len(ro[1]) >= 2
        for src in (ref, got):
            d, sp, sd = FE(symbol_clock=mode).slice(src)
            assert np.array_equal(d, ro[0]) and np.array_equal(sp, ro[1]) and np.array_equal(sd, ro[2].astype(np.uint64))
        assert np.array_equal(FE(symbol_clock=mode).run_cf32(iq), ro[0])


@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_against_the_independent_batch_model(spec, FE, mode):
    """The HIP path against tests/spec_model.py directly (the whole-array numpy restatement of SPEC 3.1 - 3.8b that the CPU
    suite pins the oracle with): baseband bit for bit from cf32 and from u8, dibits / sync positions / sync dibit indices
    with lock drops, fused and through the streaming calls -- no oracle in between."""
    import torch
    import spec_model
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, _, _ = c4fm.synth(0.8, seed=77, snr_db=14.0, frame_dibits=300, clock_ppm=120.0 if mode else 0.0, timing_offset=17)
    iq = iq[:len(iq) // 8 * 8]
    m = spec_model.Model(spec)
    fe = FE(symbol_clock=mode)
    ref_bb = spec_model.demod(spec, iq=iq)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    bb, nb = fe.demod_dev(t)
    assert nb == len(ref_bb) and np.array_equal(bits(bb[0, :nb].cpu().numpy()), bits(ref_bb))
    u8 = c4fm.to_u8(iq)
    ref_u8 = spec_model.demod(spec, u8=u8)
    got_u8 = np.concatenate([fe.demod_u8(u8[o:o + 32768]) for o in range(0, len(u8), 32768)])
    assert np.array_equal(bits(got_u8), bits(ref_u8))
    drops = [5000, 5001, 20011, 30000]
    recv = (lambda b, d: m.receive_tracking(b, spec, d)) if mode else m.receive
    for rs in ([], drops):
        ref = recv(ref_bb, rs)
        assert len(ref[1]) >= 8
        if rs:
            fe.resync_at_dev(torch.tensor(rs, dtype=torch.int64, device="cuda"))
        dib, res, sp, sd = fe.slice_dev(bb[:, :nb].contiguous(), nb, sync_cap=256)
        r = parse_results(res)[0]
        nd, ns = int(r["n_dibits"]), int(r["n_sync"])
        assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])
        assert ns == len(ref[1]) and np.array_equal(sp[0, :ns].cpu().numpy(), ref[1])
        assert np.array_equal(sd[0, :ns].cpu().numpy().astype(np.uint64), ref[2])
        if rs:
            fe.resync_at_dev(torch.tensor(rs, dtype=torch.int64, device="cuda"))
        d2, r2 = fe.run_dev(t)
        n2 = int(parse_results(r2)[0]["n_dibits"])
        assert n2 == len(ref[0]) and np.array_equal(d2[0, :n2].cpu().numpy(), ref[0])
    fe2 = FE(symbol_clock=mode)
    free = recv(ref_bb, [])
    got = np.concatenate([fe2.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
    assert np.array_equal(got, free[0])


def test_nid_after_sync_matches_oracle(O, FE):
    """Next row: NID after every frame sync, GPU exhaustive BCH(63,16,23) search == oracle, record
    for record, on clean frames, on frames with injected bit errors (<= 11 corrected, more rejected) and at the
    truncated end of the stream."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd._lib import NID_DTYPE
    from p25rx_amd.frontend import parse_results
    nidf = lambda f: ((0x100 + 37 * f) & 0xFFF, (5 * f + 1) & 15)
    iq, truth, _ = c4fm.synth(6.0, seed=44, snr_db=18.0, frame_dibits=400, nid=nidf)
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    bb, nb = fe.demod_dev(t)
    dib, res, sp, sd = fe.slice_dev(bb[0], nb, sync_cap=256)
    r = parse_results(res)[0]
    nd, ns = int(r["n_dibits"]), int(r["n_sync"])
    assert ns >= 70
    host_dib = dib[0, :nd].cpu().numpy().copy()
    rng = np.random.default_rng(1)
    sdh = sd[0, :ns].cpu().numpy().astype(np.uint64)
    for k in range(ns):                                               # inject k % 15 bit errors into NID k
        pos = [j for j in range(33) if j != 11]
        for j in rng.choice(pos, size=k % 15, replace=False):
            if int(sdh[k]) + j < nd:
                host_dib[int(sdh[k]) + j] ^= 1 << int(rng.integers(0, 2))
    d2 = torch.from_numpy(host_dib).cuda()
    got = np.frombuffer(fe.nid_dev(d2, nd, sd[0, :ns], sp[0, :ns]).cpu().numpy().tobytes(), dtype=NID_DTYPE)
    ref = O.nid_decode(host_dib, sdh, sp[0, :ns].cpu().numpy())
    assert got.tobytes() == ref.tobytes()
    ok = got[got["valid"] == 1]
    assert len(ok) >= ns * 11 // 15 - 2
    frames = (ok["sync_pos"] - int(got["sync_pos"][0])) // 4000       # 400 dibits per frame = 4000 baseband samples
    assert all((int(n["nac"]), int(n["duid"])) == nidf(int(f)) for n, f in zip(ok, frames))
    # truncated stream: the last NID is cut
    cut = int(sdh[-1]) + 10
    g2 = np.frombuffer(fe.nid_dev(d2, cut, sd[0, :ns], sp[0, :ns]).cpu().numpy().tobytes(), dtype=NID_DTYPE)
    assert g2["valid"][-1] == -1 and g2[:-1].tobytes() == got[:-1].tobytes()


def test_channel_batch_nid_and_stats(O, FE):
    """Next row: per-channel observability for a channel batch -- sigPower (src/demod.rs:95-101)
    and the "bch" CodeStats row (src/hub.rs:559, 574-581) -- with no host round trip between the kernels.  Channels
    at different SNRs (one noise only, one so noisy that NIDs fail) against the oracle run channel by channel."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd._lib import NID_DTYPE, CHAN_STATS_DTYPE
    from p25rx_amd.frontend import parse_results
    snrs = [30.0, 12.0, -4.0, None, -9.0]         # dB in 240 kHz: below ~0 dB the discriminator clicks corrupt NID bits
    C, secs, cap = len(snrs), 3.0, 64
    nidf = lambda f: ((0x293 + f) & 0xFFF, (3 * f) & 15)
    n = int(secs * 240000)
    iq = np.zeros((C, n), dtype=np.complex64)
    rng = np.random.default_rng(9)
    for c, snr in enumerate(snrs):
        if snr is None:
            iq[c] = (0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        else:
            iq[c] = c4fm.synth(secs, seed=70 + c, snr_db=snr, frame_dibits=600, nid=nidf, amplitude=0.2 + 0.1 * c)[0]
    t = torch.from_numpy(iq.view(np.float32).reshape(C, n, 2)).cuda()
    fe = FE(n_channels=C)
    bb, nb, pw = fe.demod_dev(t, want_power=True)
    dib, res, sp, sd = fe.slice_dev(bb, nb, sync_cap=cap)
    nid = fe.nid_batch_dev(dib, res, sd, sp)
    st = np.frombuffer(fe.chan_stats_dev(res, nid, pw).cpu().numpy().tobytes(), dtype=CHAN_STATS_DTYPE)
    rs = parse_results(res)
    nid_h = nid.cpu().numpy()
    seen_err = seen_fixed = False
    for c in range(C):
        d = O.Demod()
        bbo, p_ref = d.feed_cf32(iq[c], want_power=True)
        r = O.Recv()
        dd, spo, sdo = r.feed(bbo)
        ns = len(spo)
        assert int(rs[c]["n_sync"]) == ns and int(rs[c]["n_dibits"]) == len(dd) and ns <= cap
        got = np.frombuffer(nid_h[c, :ns].tobytes(), dtype=NID_DTYPE)
        ref = O.nid_decode(dd, np.asarray(sdo, dtype=np.uint64), np.asarray(spo, dtype=np.int64))
        assert got.tobytes() == ref.tobytes(), c
        assert not nid_h[c, ns:].any()                              # rows past n_sync untouched
        words = int((ref["valid"] >= 0).sum()); errs = int((ref["valid"] == 0).sum())
        fixed = int(ref["n_errors"][ref["valid"] == 1].sum())
        s = st[c]
        assert (int(s["bch"]["words"]), int(s["bch"]["errs"]), int(s["bch"]["fixed"]), int(s["bch"]["size"])) == (words, errs, fixed, 63)
        assert int(s["n_sync"]) == ns and int(s["n_dibits"]) == len(dd)
        assert int(s["locked"]) == (1 if ns else 0)
        assert int(s["last_sync_pos"]) == (int(spo[-1]) if ns else -1)
        assert abs(float(s["sig_power_dbm"]) - p_ref) <= 1e-3       # dB; tree vs sequential reduction
        seen_err |= errs > 0
        seen_fixed |= fixed > 0
    assert int(st[3]["n_sync"]) == 0 and int(st[0]["bch"]["errs"]) == 0 and int(st[0]["bch"]["words"]) >= 11
    assert seen_fixed and seen_err
    st2 = np.frombuffer(fe.chan_stats_dev(res).cpu().numpy().tobytes(), dtype=CHAN_STATS_DTYPE)
    assert np.isnan(st2["sig_power_dbm"]).all() and not st2["bch"]["words"].any()


def test_channeliser_parity_and_end_to_end(O, FE):
    """Next row: polyphase channeliser.  (i) GPU factored-DFT form vs the oracle's plain mix /
    filter / decimate in fp64, tolerance 2e-6 * sum|h| * max|x| (fp32 rounding of ~10 operations deep; floating-point
    stage, no reference numbers); whole capture and ranges with history on odd grids; channel 0 vs K0.
    (ii) end to end: three C4FM carriers on raster slots of one 2.4 Msps capture -> channeliser -> the 192-channel
    front end: each carrier's dibits equal the generator's truth, idle slots never lock."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    spec = O.load_spec()
    hsum = float(np.abs(np.array(spec["pre_taps"], dtype=np.float64)).sum())
    rng = np.random.default_rng(21)
    n = 30008
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
    tol = 2e-6 * hsum * float(np.abs(x).max())
    ref = O.channelise(x)
    fe = FE()
    t = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()

    def host(y, no):
        return y[:, :no].cpu().numpy().view(np.complex64)[..., 0]
    y, no = fe.channelise_dev(t)
    assert no == ref.shape[1]
    assert np.abs(host(y, no) - ref).max() <= tol
    p, _ = fe.predecim_dev(t)
    assert np.abs(p[0, :no].cpu().numpy().view(np.complex64)[:, 0] - host(y, no)[0]).max() <= tol
    for off, n_hist in ((10000, 80), (12346, 96), (20002, 20002), (30000, 200)):   # 16-B aligned range starts
        y2, no2 = fe.channelise_dev(t, n_hist=n_hist, abs0=off, offset=off)
        first = len([m for m in range(ref.shape[1]) if 10 * m + 9 < off])
        assert no2 == ref.shape[1] - first
        assert no2 == 0 or np.abs(host(y2, no2) - ref[:, first:]).max() <= tol, off
    with pytest.raises(Exception):
        fe.channelise_dev(t, n_hist=0, abs0=1, offset=1)               # misaligned pointer is rejected

    carriers = {5: (31, 1.0), 100: (32, 0.6), 190: (33, 0.8)}
    wide, truth = c4fm.synth_wideband(0.5, carriers, snr_db=22.0, seed=3)
    tw = torch.from_numpy(wide.view(np.float32).reshape(-1, 2)).cuda()
    yc, nc = fe.channelise_dev(tw)
    fe192 = FE(n_channels=192)
    dib, res = fe192.run_dev(yc[:, :nc])
    r = parse_results(res)
    for c in range(192):
        nd = int(r["n_dibits"][c])
        if c in carriers:
            got = dib[c, :nd].cpu().numpy()
            k = min(len(got), len(truth[c]) - 24)
            assert k > 2000 and np.array_equal(got[:k], truth[c][24:24 + k]), c
            chan = yc[c, :nc].cpu().numpy().view(np.complex64)[:, 0]
            assert np.array_equal(got, O.run_cf32(chan)), c             # downstream of the channeliser: bit-exact as ever
        elif min(abs(c - k) if abs(c - k) <= 96 else 192 - abs(c - k) for k in carriers) > 1:
            assert int(r["n_sync"][c]) == 0, c                          # idle slot (adjacent slots see the skirt)


def test_dense_sync_events_and_noise(O, FE):
    """Symbol receiver stress: sync words so dense that a 7680-sample tile holds dozens of detections (K2's sorted
    in-tile list and counts, K4's per-detection segments and recomputed thresholds), timing jumps between frames, loud noise that
    produces candidates but no lock, and an amplitude step -- dibits, sync positions and sync dibit indices equal the
    oracle's, through the one-shot device path and through ragged streaming chunks."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    rng = np.random.default_rng(77)
    pieces = []
    for fd, snr, amp, toff in ((26, 25.0, 0.5, 0), (40, 14.0, 0.25, 3), (100, 10.0, 0.7, 17), (31, 30.0, 0.1, 41)):
        iq = c4fm.synth(0.6, seed=200 + fd, snr_db=snr, frame_dibits=fd, amplitude=amp, timing_offset=toff)[0]
        pieces.append(iq)
        pieces.append((0.3 * (rng.standard_normal(5003) + 1j * rng.standard_normal(5003))).astype(np.complex64))
    iq = np.concatenate(pieces)
    iq = iq[:len(iq) // 8 * 8]
    bb_ref = O.Demod().feed_cf32(iq)
    rcv = O.Recv()
    d_ref, sp_ref, sd_ref = rcv.feed(bb_ref)
    assert len(sp_ref) > 250                                         # several events per tile on the dense stretches
    gaps = np.diff(sp_ref)
    assert (gaps < 512).sum() > 100
    fe = FE()
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    bb, nb = fe.demod_dev(t)
    assert np.array_equal(bb[0, :nb].cpu().numpy().view(np.uint32), bb_ref.view(np.uint32))
    dib, res, sp, sd = fe.slice_dev(bb[0], nb, sync_cap=4096)
    r = parse_results(res)[0]
    nd, ns = int(r["n_dibits"]), int(r["n_sync"])
    assert nd == len(d_ref) and ns == len(sp_ref)
    assert np.array_equal(dib[0, :nd].cpu().numpy(), d_ref)
    assert np.array_equal(sp[0, :ns].cpu().numpy(), sp_ref)
    assert np.array_equal(sd[0, :ns].cpu().numpy().astype(np.uint64), sd_ref.astype(np.uint64))
    # streaming, ragged chunks (host entry points)
    fe2 = FE()
    cuts = sorted(set(int(x) for x in rng.integers(1, nb - 1, size=40))) + [nb]
    got_d, got_s, o = [], [], 0
    for c in cuts:
        dd, ss, _ = fe2.slice(bb_ref[o:c])
        got_d.append(dd); got_s.append(ss); o = c
    assert np.array_equal(np.concatenate(got_d), d_ref)
    assert np.array_equal(np.concatenate(got_s), sp_ref)


def test_config2_full_size_properties(FE):
    """BASELINE.json configs[1] at its FULL size (1 channel x 600 s = 1.44e8 samples, generated in HBM), checked through
    size-independent properties: demod(mod(d)) == d for the cf32 capture and for the same capture quantised to RTL-SDR
    u8 pairs; sync count = frames; and range invariance -- the capture processed as two device ranges with history and
    an anchor hand-over gives the same baseband bits and the same dibits as the single pass."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    n = 600 * 240000
    iq, truth = c4fm.synth_torch(n, seed=1000, device="cuda", snr_db=30.0)
    fe = FE()
    dib, res = fe.run_dev(iq)
    r = parse_results(res)[0]
    nd = int(r["n_dibits"])
    got = dib[0, :nd].cpu().numpy()
    k = min(nd, len(truth) - 24)
    assert k > 2879000 and np.array_equal(got[:k], truth[24:24 + k])
    assert int(r["n_sync"]) == len(truth) // 864 + (1 if len(truth) % 864 >= 24 else 0)
    # u8 quantisation of the same capture (inverse of SPEC 3.1): still every symbol
    u8 = torch.clamp(torch.round((iq + 1.0) * 127.5), 0, 255).to(torch.uint8)
    d8, r8 = fe.run_dev(u8)
    g8 = d8[0, :int(parse_results(r8)[0]["n_dibits"])].cpu().numpy()
    k8 = min(len(g8), len(truth) - 24)
    assert k8 > 2879000 and np.array_equal(g8[:k8], truth[24:24 + k8])
    del u8, d8
    # two ranges: [0, cut) and [cut, n) with 2000 samples of history; baseband bit-identical, dibits identical
    cut = 71_000_008
    bb_all, nb_all = fe.demod_dev(iq)
    bb1, nb1 = fe.demod_dev(iq[:cut])
    bb2, nb2 = fe.demod_dev(iq[cut - 2000:], n_hist=2000, abs0=cut, offset=2000)
    assert nb1 + nb2 == nb_all
    assert torch.equal(bb1[0, :nb1].view(torch.int32), bb_all[0, :nb1].view(torch.int32))
    assert torch.equal(bb2[0, :nb2].view(torch.int32), bb_all[0, nb1:nb_all].view(torch.int32))
    d1, r1, _, _ = fe.slice_dev(bb_all[0], nb1)
    a1 = parse_results(r1)[0]
    d2, r2, _, _ = fe.slice_dev(bb_all[0], nb2, n_hist_bb=nb1, abs_bb0=nb1, anchor_in=a1["anchor_out"], offset=nb1)
    n1, n2 = int(a1["n_dibits"]), int(parse_results(r2)[0]["n_dibits"])
    assert n1 + n2 == nd
    assert torch.equal(d1[0, :n1], dib[0, :n1]) and torch.equal(d2[0, :n2], dib[0, n1:nd])


def test_config4_full_size_properties(FE):
    """BASELINE.json configs[3] at its FULL size: 256 channels x 60 s x 240 ksps, channel-major (29.5 GB, 3.7e9 samples --
    beyond 32-bit indices).  Channel c is the same generated capture delayed by c symbols, so every channel's dibits
    must equal the modulator's symbols at its own offset, and all channels must agree with each other."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    C, n = 256, 60 * 240000
    base, truth = c4fm.synth_torch(n + 50 * C, seed=4242, device="cuda", snr_db=30.0)
    iq = torch.empty((C, n, 2), dtype=torch.float32, device="cuda")
    for c in range(C):
        iq[c] = base[50 * c: 50 * c + n]
    del base
    fe = FE(n_channels=C)
    dib, res = fe.run_dev(iq)
    r = parse_results(res)
    dib_h = dib.cpu().numpy()
    for c in range(C):
        nd = int(r["n_dibits"][c])
        # the first sync word that lies fully inside channel c's capture is frame 0's (4 lead symbols precede it) or frame 1's
        ok = False
        for j in (24, 864 + 24):
            k = min(nd, len(truth) - j) - 8
            ok = ok or (k > 286000 and np.array_equal(dib_h[c, :k], truth[j:j + k]))
        assert ok, (c, nd)
        assert int(r["n_sync"][c]) >= 332, (c, int(r["n_sync"][c]))


def test_config5_full_size_single_gpu(FE):
    """BASELINE.json configs[4]'s capture at its FULL size on one GPU: 3 600 s x 240 ksps = 8.64e8 samples (6.9 GB: byte
    offsets beyond 32 bits inside one channel, 843 750 tiles, 412 scan groups): every one of the 17.28 M dibits equals
    the modulator's symbol, one sync per frame."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    n = 3600 * 240000
    iq, truth = c4fm.synth_torch(n, seed=77, device="cuda", snr_db=30.0)
    dib, res = FE().run_dev(iq)
    r = parse_results(res)[0]
    nd = int(r["n_dibits"])
    got = dib[0, :nd].cpu().numpy()
    k = min(nd, len(truth) - 24)
    assert k > 17_279_000 and np.array_equal(got[:k], truth[24:24 + k])
    assert int(r["n_sync"]) == len(truth) // 864 + (1 if len(truth) % 864 >= 24 else 0)


@pytest.mark.gpu
def test_run_dev_every_symbol_phase(O, FE):
    """The fused path keeps the baseband in ten polyphase planes and a locked receiver reads only the plane of its
    symbol phase: sweep the timing offset so that every plane (and its sign-bit plane) carries the sync words and the
    symbol instants at least once.  (A stale sign-bit word of plane 0 was invisible to every single-phase test.)"""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    fe = FE()
    phases = set()
    for off in range(0, 50, 3):
        iq, _, _ = c4fm.synth(0.4, seed=900 + off, snr_db=25.0, timing_offset=off, frame_dibits=200 + off)
        rx = O.Recv()
        ref, spos, _ = rx.feed(O.Demod().feed_cf32(iq))
        assert len(spos) >= 3
        phases.add(int(spos[0]) % 10)
        t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
        dib, res = fe.run_dev(t)
        r = parse_results(res)[0]
        assert int(r["n_dibits"]) == len(ref) and int(r["n_sync"]) == len(spos), off
        assert np.array_equal(dib[0, :len(ref)].cpu().numpy(), ref), off
    assert len(phases) == 10, phases


@pytest.mark.gpu
def test_config5_full_size_eight_shards_one_gpu(FE):
    """BASELINE.json configs[4] at FULL size through the sharded path: the 8.64e8-sample capture as 8 contiguous
    shards (the code path of 8 ranks, run back to back on one GPU): pass 1 in its two launches (main, then the head
    that needs the halo), device resolve, pass 2, dibit compaction.  Shards 1..7 start at baseband indices up to
    1.5e8 (64-bit absolute indices, negative local ones in the head); the compacted stream must equal the single
    pass over the whole capture, byte for byte."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import parse_results, n_baseband
    n, W = 3600 * 240000, 8
    iq, truth = c4fm.synth_torch(n, seed=78, device="cuda", snr_db=30.0)
    ref, rres = FE().run_dev(iq)
    n_ref = int(parse_results(rres)[0]["n_dibits"])
    per = n // W
    assert per % 8 == 0
    fes = [FE() for _ in range(W)]
    halo = fes[0].shard_halo()
    summ_all = torch.empty((W, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    bb0 = [n_baseband(0, r * per) for r in range(W)]
    bbn = [n_baseband(r * per, per) for r in range(W)]
    cap = (max(bbn) // 10 + 64 + 15) // 16 * 16
    for r in range(W):
        h = halo if r else 0
        view = iq[r * per - h:(r + 1) * per]
        fes[r].shard_pass1_main(view, offset=h, n_hist=h, abs0=r * per)
        res = fes[r].shard_pass1_finish(view, offset=h, n_hist=h, abs0=r * per)
        summ_all[r] = res[0]
    d_anc, d_off = fes[0].shard_resolve_dev(summ_all, torch.tensor(bb0, dtype=torch.int64, device="cuda"),
                                            torch.tensor(bbn, dtype=torch.int64, device="cuda"))
    gathered = torch.zeros((W, cap), dtype=torch.uint8, device="cuda")
    for r in range(W):
        fes[r].shard_pass2(d_anc[r:r + 1], bbn[r], iq.device, dibits=gathered[r:r + 1])
    off = d_off.cpu().numpy()
    assert int(off[-1]) == n_ref and all(off[r + 1] - off[r] <= cap for r in range(W))
    stream = torch.zeros(n_ref + 16, dtype=torch.uint8, device="cuda")
    fes[0].shard_compact_dev(gathered, d_off, stream)
    assert torch.equal(stream[:n_ref], ref[0, :n_ref])
    got = stream[:n_ref].cpu().numpy()
    k = min(n_ref, len(truth) - 24)
    assert k > 17_279_000 and np.array_equal(got[:k], truth[24:24 + k])
    # a pass 2 after another call touched the handle must refuse instead of slicing stale scratch (p25fe.h, ORDERING)
    fes[1].run_dev(iq[:240000])
    with pytest.raises(Exception):
        fes[1].shard_pass2(d_anc[1:2], bbn[1], iq.device)


@pytest.mark.gpu
def test_config3_full_size_end_to_end(FE):
    """BASELINE.json configs[2] end to end at full size: 2.4 Msps x 60 s = 1.44e8 samples -> stage 0 (80-tap 10:1
    pre-decimator) -> K1..K4 -> dibits, with demod(mod(d)) == d for all 288 000 symbols.  The wideband capture is the
    240 ksps C4FM signal held for ten samples each (zero-order hold: its images sit in the nulls of the hold and in
    the stop band of the pre-decimator)."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    n240 = 60 * 240000
    iq240, truth = c4fm.synth_torch(n240, seed=31, device="cuda", snr_db=30.0)
    wide = iq240.repeat_interleave(10, dim=0).contiguous()                          # [1.44e8, 2] @ 2.4 Msps
    assert wide.shape[0] == 144_000_000
    fe = FE()
    narrow, n_out = fe.predecim_dev(wide)
    assert n_out == n240
    dib, res = fe.run_dev(narrow[:, :n_out])
    r = parse_results(res)[0]
    nd = int(r["n_dibits"])
    got = dib[0, :nd].cpu().numpy()
    kk, jj, n_match, n_err = c4fm.align_dibits(got, truth)
    assert n_err == 0 and n_match > 287_000, (kk, jj, n_match, n_err)
    assert int(r["n_sync"]) >= len(truth) // 864


@pytest.mark.gpu
def test_sync_screen_keeps_candidates_with_wrong_sign_symbols(O, FE):
    """K2 evaluates the exact correlation only where at most 4 of the 24 symbol-spaced samples have the wrong SIGN
    (a necessary condition for c^2 >= 20.4 e, p25fe_recv.hip).  Build basebands whose sync words carry 1, 2 and 3
    symbols of tiny wrong-sign amplitude (c^2 / e = 23, 22, 21: still candidates) and 4 (c^2 / e = 20 < 20.4: not one),
    at every symbol phase, next to exact zeros and negative zeros; detections and dibits must equal the oracle's."""
    spec = O.load_spec()
    mask = spec["sync_sign_mask"]
    rng = np.random.default_rng(5)
    n = 40000
    bb = (rng.standard_normal(n) * 0.02).astype(np.float32)
    bb[1000:1100] = 0.0
    bb[1100:1200] = -0.0
    expect = []
    pos = 3000
    for flips in (0, 1, 2, 3, 4, 3, 2, 1):
        for phase in range(10):
            s_last = pos + phase
            bad = set(rng.choice(24, size=flips, replace=False).tolist())
            for j in range(24):
                g = 1.0 if (mask >> j) & 1 else -1.0
                idx = s_last - 10 * (23 - j)
                bb[idx - 4:idx + 5] = 0.0                                  # isolate the symbol so that the peak is at s_last
                bb[idx] = np.float32(-1e-4 * g if j in bad else 0.3 * g)
            expect.append(flips)
            pos += 420
    assert pos < n - 500
    dib, spos, sdib = O.Recv().feed(bb)
    # the construction does what it says: every word with <= 3 flipped symbols is detected by the oracle
    assert len(spos) >= sum(1 for f in expect if f <= 3)
    got = FE().slice(bb)
    assert np.array_equal(got[1], spos) and np.array_equal(got[2], sdib)
    assert np.array_equal(got[0], dib)


@pytest.mark.gpu
def test_run_dev_lengths_around_layout_boundaries(O, FE):
    """The fused path's geometry has several granules: 880-output K1 segments, 320-output sub-tiles, 32-symbol blocks of
    the polyphase layout, 7680-sample receiver tiles.  Capture lengths that end just before / on / after each of them
    (and a few random ones), cf32 and u8: dibit count, dibits and detections equal the oracle's."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq_all, _, _ = c4fm.synth(1.2, seed=41, snr_db=20.0, frame_dibits=230, timing_offset=7)
    u8_all = c4fm.to_u8(iq_all)
    rng = np.random.default_rng(9)
    lens = set()
    for base_bb in (880, 320 * 3, 7680, 7680 * 2, 320 * 24, 880 * 11, 10 * 32 * 7):
        for d in (-6, -5, -1, 0, 1, 4, 5, 11):
            lens.add(5 * base_bb + d)
    lens |= set(int(x) for x in rng.integers(2000, len(iq_all) - 8, size=12))
    fe, fe8 = FE(), FE()
    for n in sorted(lens):
        n = max(8, min(n, len(iq_all)))
        iq = iq_all[:n]
        ref = O.run_cf32(iq)
        t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
        dib, res = fe.run_dev(t)
        r = parse_results(res)[0]
        assert int(r["n_dibits"]) == len(ref), n
        assert np.array_equal(dib[0, :len(ref)].cpu().numpy(), ref), n
        if n % 97 < 40:                                          # a subset through the u8 kernel as well
            ref8 = O.Recv().feed(O.Demod().feed_u8(u8_all[:2 * n]))[0]
            t8 = torch.from_numpy(u8_all[:2 * n].reshape(-1, 2)).cuda()
            d8, r8 = fe8.run_dev(t8)
            assert int(parse_results(r8)[0]["n_dibits"]) == len(ref8), n
            assert np.array_equal(d8[0, :len(ref8)].cpu().numpy(), ref8), n


@pytest.mark.gpu
def test_run_dev_pipelined_sequence(O, FE):
    """p25fe_run_dev_pipelined: a sequence of DIFFERENT captures (lengths from empty to 3 s, cf32 and u8) enqueued back to
    back on one handle -- K1 of call i + 1 overlaps the receive kernels of call i, the scratch sets alternate -- gives,
    after p25fe_join_dev, exactly the oracle's dibits for every capture; a p25fe_run_dev in the middle of the sequence
    (it joins first) and a second round over the same handle do too."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    caps, refs = [], []
    for k, (secs, u8) in enumerate(((3.0, False), (0.25, False), (0.00004, False), (1.7, True), (0.9, False), (2.2, True), (0.05, False))):
        iq = c4fm.synth(max(secs, 0.01), seed=300 + k, snr_db=18.0 + k, frame_dibits=150 + 31 * k, timing_offset=3 * k)[0]
        iq = iq[:int(secs * 240000) // 8 * 8]
        if u8:
            x = c4fm.to_u8(iq)
            refs.append(O.Recv().feed(O.Demod().feed_u8(x))[0] if len(x) else np.zeros(0, np.uint8))
            caps.append(torch.from_numpy(x.reshape(-1, 2)).cuda())
        else:
            refs.append(O.run_cf32(iq) if len(iq) else np.zeros(0, np.uint8))
            caps.append(torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda())
    fe = FE()
    for rnd in range(2):
        outs = [fe.run_dev_pipelined(t) for t in caps]
        fe.join_dev()
        torch.cuda.synchronize()
        for k, ((dib, res), ref) in enumerate(zip(outs, refs)):
            r = parse_results(res)[0]
            assert int(r["n_dibits"]) == len(ref), (rnd, k)
            assert np.array_equal(dib[0, :len(ref)].cpu().numpy(), ref), (rnd, k)
        # an ordinary call between pipelined ones joins by itself
        a = fe.run_dev_pipelined(caps[0])
        b = fe.run_dev(caps[4])
        c = fe.run_dev_pipelined(caps[3])
        fe.join_dev()
        torch.cuda.synchronize()
        for (dib, res), ref in ((a, refs[0]), (b, refs[4]), (c, refs[3])):
            assert int(parse_results(res)[0]["n_dibits"]) == len(ref)
            assert np.array_equal(dib[0, :len(ref)].cpu().numpy(), ref)


@pytest.mark.gpu
def test_run_dev_pipelined_channel_batches(O, FE):
    """The pipelined entry point on a 4-channel handle: two different batches (different lengths: the scratch is resized
    between calls) alternate three times; every channel of every call equals the oracle's dibits."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    C = 4
    batches, refs = [], []
    for b, secs in enumerate((0.8, 1.3)):
        chans = [c4fm.synth(secs, seed=500 + 10 * b + c, snr_db=16.0 + 2 * c, frame_dibits=120 + 40 * c, timing_offset=c)[0] for c in range(C)]
        n = min(len(x) for x in chans) // 8 * 8
        arr = np.stack([x[:n] for x in chans])
        refs.append([O.run_cf32(arr[c]) for c in range(C)])
        batches.append(torch.from_numpy(arr.view(np.float32).reshape(C, n, 2)).cuda())
    fe = FE(n_channels=C)
    outs = []
    for rnd in range(3):
        for b in (0, 1):
            outs.append((b, fe.run_dev_pipelined(batches[b])))
    fe.join_dev()
    torch.cuda.synchronize()
    for b, (dib, res) in outs:
        r = parse_results(res)
        for c in range(C):
            nd = int(r[c]["n_dibits"])
            assert nd == len(refs[b][c]), (b, c)
            assert np.array_equal(dib[c, :nd].cpu().numpy(), refs[b][c]), (b, c)


@pytest.mark.gpu
def test_profile_hook_levels(FE):
    """p25fe_profile_enable: level 1 records every kernel of every call, level 2 only K1, level 3 only K1 and only on
    every 8th call; p25fe_profile_read reports how many of the kept slots ran K1."""
    import torch
    from p25rx_amd import c4fm
    iq = c4fm.synth(0.5, seed=9, snr_db=20.0)[0]
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    for level, calls, want in ((1, 5, 5), (2, 6, 6), (3, 32, 4), (3, 9, 2)):
        fe.profile_enable(level)
        for _ in range(calls):
            fe.run_dev_pipelined(t) if level == 3 else fe.run_dev(t)
        fe.join_dev()
        torch.cuda.synchronize()
        ms, n = fe.profile_read()
        assert n == want, (level, calls, n)
        assert ms[0] > 0.0
        assert (ms[1] > 0.0 and ms[3] > 0.0) if level == 1 else (ms[1] == 0.0 and ms[2] == 0.0 and ms[3] == 0.0)
    fe.profile_enable(False)


def test_head_that_never_arrives_times_out_instead_of_hanging(O, FE):
    """the detection's first tiles poll the flag the head launch's last workgroup writes (p25fe_shard_pass1_head on another
    stream: in the N > 1 step that stream sits behind the halo's ncclRecv).  If the head does not come -- here: its stream is held up
    by a 5 s spin kernel in front of it -- the poll gives up after 2 s of device wall clock instead of spinning for ever, the queue
    drains, and p25fe_shard_head_check (p25fe_shard_offsets in the RCCL step) reports P25FE_ERR_TIMEOUT.  The same procedure with the
    head delivered in time reports OK and the plain pass 1's record."""
    import time
    import torch
    from p25rx_amd import c4fm, _lib
    from p25rx_amd.frontend import parse_results
    iq, _, _ = c4fm.synth(1.0, seed=43, snr_db=20.0, frame_dibits=700)
    t_all = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    cut = 120008
    halo = FE().shard_halo()
    t = t_all[cut - halo:]
    side = torch.cuda.Stream()
    # a kernel that keeps a stream busy for ~5 s: torch's spin kernel, calibrated here
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(100_000_000)
    torch.cuda.synchronize()
    per_cycle = (time.perf_counter() - t0) / 1e8
    hold = int(5.0 / per_cycle)
    # warm-up of everything that is not the wait (allocations, first launches), head delivered: OK
    fe = FE()
    fe.shard_pass1_main(t, offset=halo, n_hist=halo, abs0=cut)
    with torch.cuda.stream(side):
        fe.shard_pass1_head(t, offset=halo, n_hist=halo, abs0=cut)
    res = fe.shard_pass1_finish(t, offset=halo, n_hist=halo, abs0=cut)
    torch.cuda.synchronize()
    fe.shard_head_check()
    ref = FE().shard_pass1(t, offset=halo, n_hist=halo, abs0=cut)
    torch.cuda.synchronize()
    assert parse_results(res)[0].tobytes() == parse_results(ref)[0].tobytes()
    # the head held up for 5 s: the detection gives up after 2
    fe.shard_pass1_main(t, offset=halo, n_hist=halo, abs0=cut)
    with torch.cuda.stream(side):
        torch.cuda._sleep(hold)
        fe.shard_pass1_head(t, offset=halo, n_hist=halo, abs0=cut)
    t0 = time.perf_counter()
    fe.shard_pass1_finish(t, offset=halo, n_hist=halo, abs0=cut)     # detection: its first tiles wait for the head's flag
    torch.cuda.current_stream().synchronize()                        # returns: the wait is bounded
    waited = time.perf_counter() - t0
    assert 1.5 < waited < 4.5, waited
    torch.cuda.synchronize()                                         # (the head arrives, too late)
    with pytest.raises(_lib.P25feError) as e:
        fe.shard_head_check()
    assert e.value.status == _lib.ERR_TIMEOUT
