"""GPU parity tests of the general symbol receiver: the tracking symbol clock (docs/SPEC.md 3.8b -- north_star's
"symbol-clock interpolator") and lock drops INSIDE device-resident ranges (MessageReceiver::resync, src/recv.rs:136, 179,
at given sample indices).  Everything is compared with the CPU oracle bit for bit: dibits, sync positions, sync dibit
indices."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def FE():
    from p25rx_amd.frontend import FrontEnd
    return FrontEnd


def oracle_recv(O, bb, mode, resync=()):
    """The oracle's receiver over bb with resync() called before each listed sample index."""
    r = O.Recv(O.make_config(symbol_clock=mode))
    outs, o = [], 0
    for q in sorted(resync):
        q = min(max(int(q), 0), len(bb))
        outs.append(r.feed(bb[o:q]))
        r.resync()
        o = q
    outs.append(r.feed(bb[o:]))
    return (np.concatenate([x[0] for x in outs]), np.concatenate([x[1] for x in outs]),
            np.concatenate([x[2] for x in outs]).astype(np.uint64))          # (sync_dibit already counts the receiver's total)


def dev_slice(fe, bb, resync=None, sync_cap=4096):
    import torch
    from p25rx_amd.frontend import parse_results
    t = torch.from_numpy(np.ascontiguousarray(bb)).cuda()
    if resync is not None:
        fe.resync_at_dev(torch.tensor(sorted(int(q) for q in resync), dtype=torch.int64, device="cuda"))
    dib, res, sp, sd = fe.slice_dev(t, len(bb), sync_cap=sync_cap)
    r = parse_results(res)[0]
    nd, ns = int(r["n_dibits"]), int(r["n_sync"])
    return dib[0, :nd].cpu().numpy(), sp[0, :ns].cpu().numpy(), sd[0, :ns].cpu().numpy().astype(np.uint64), r


def same(got, ref, what=""):
    for k, name in enumerate(("dibits", "sync_pos", "sync_dibit")):
        assert len(got[k]) == len(ref[k]), "%s %s: %d vs %d" % (what, name, len(got[k]), len(ref[k]))
        assert np.array_equal(got[k], ref[k]), "%s %s differ" % (what, name)


@pytest.mark.parametrize("ppm,frame", [(100.0, 3000), (-100.0, 3000), (35.0, 8640), (0.0, 864)])
def test_tracking_clock_device_matches_oracle(O, FE, ppm, frame):
    """Mode 1 on a capture with a sample-clock error: the device receiver (linear baseband in, and the fused IQ path whose K1
    writes the planes two samples late) equals the oracle; every frame after the first decodes to the modulator's dibits."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, truth, _ = c4fm.synth(4 * frame / 4800.0 + 0.3, seed=21, snr_db=30.0, frame_dibits=frame, clock_ppm=ppm)
    iq = iq[:len(iq) // 8 * 8]
    bb = O.Demod().feed_cf32(iq)
    ref = oracle_recv(O, bb, 1)
    assert len(ref[1]) >= 5
    fe = FE(symbol_clock=1)
    got = dev_slice(fe, bb)
    same(got, ref, "slice_dev")
    for k in range(1, len(ref[2]) - 1):                              # locked frames are error free
        assert np.array_equal(ref[0][int(ref[2][k]):int(ref[2][k + 1])], truth[frame * k + 24:frame * (k + 1) + 24])
    r = got[3]
    # the summary a time shard hands to p25fe_shard_resolve: the first detection's governed interval ends at the NEXT event of
    # the range, not at the end of its 7 680-sample tile (sync words are further apart than a tile here)
    assert int(r["first_event"]) == int(ref[1][0]) + 5 and int(r["first_seg_end"]) == int(ref[1][1]) + 6
    a = r["anchor_out"]
    # (the period is kept in quarter samples over four times the symbol count: docs/SPEC.md 3.8b)
    if ppm:
        assert int(a["period_n"]) == 4 * frame and abs(int(a["period_d"]) - 4 * frame * 10 * (1 + ppm * 1e-6)) <= 6
    else:
        assert int(a["period_n"]) in (1, 4 * frame) and abs(int(a["period_d"]) / int(a["period_n"]) - 10.0) < 0.01
    assert (int(a["valid"]) & 1) == 1 and -2 <= ((int(a["valid"]) >> 8) ^ 4) - 4 <= 2
    # fused: IQ -> dibits in one pass
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    dib, res = fe.run_dev(t)
    nd = int(parse_results(res)[0]["n_dibits"])
    assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])
    # streaming through the host entry points, ragged chunks, state (anchor + clock) carried in the handle
    fe2, rng = FE(symbol_clock=1), np.random.default_rng(3)
    parts, o = [], 0
    sizes = [1, 1, 2, 3, 239, 240, 241, 242, 243]
    while o < len(bb):
        n = sizes.pop(0) if sizes else int(rng.integers(1, 12000))
        parts.append(fe2.slice(bb[o:o + n]))
        o += n
    same([np.concatenate([p[k] for p in parts]) for k in range(3)], ref, "streaming")
    fe3 = FE(symbol_clock=1)
    got3 = np.concatenate([fe3.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
    assert np.array_equal(got3, ref[0])


@pytest.mark.parametrize("ppm,frame", [(150.0, 864), (-250.0, 864), (100.0, 3000), (0.0, 864)])
def test_reslice_device_matches_oracle(O, FE, ppm, frame):
    """SPEC 3.8c (symbol_clock = 2): the calls that hold the whole range -- p25fe_slice_dev on a linear baseband, p25fe_run_dev and
    p25fe_run_dev_pipelined from IQ -- equal the oracle's two-pass restatement bit for bit, and EVERY frame, the first of the lock run
    included, decodes to the modulator's dibits; the streaming calls keep mode 1's causal rule in this mode."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, truth, _ = c4fm.synth(5 * frame / 4800.0 + 0.3, seed=23, snr_db=30.0, frame_dibits=frame, clock_ppm=ppm)
    iq = iq[:len(iq) // 8 * 8]
    bb = O.Demod().feed_cf32(iq)
    ref = O.recv_range(bb, O.make_config(symbol_clock=2))
    one = oracle_recv(O, bb, 1)
    assert len(ref[1]) >= 5 and np.array_equal(ref[1], one[1])
    for k in range(0, len(ref[2]) - 1):                              # every frame between two sync words is error free, the first too
        assert np.array_equal(ref[0][int(ref[2][k]):int(ref[2][k + 1])], truth[frame * k + 24:frame * (k + 1) + 24]), k
    if abs(ppm) >= 150:
        assert not np.array_equal(one[0][:int(one[2][1])], truth[24:24 + int(one[2][1])])     # mode 1 does lose symbols there
    fe = FE(symbol_clock=2)
    got = dev_slice(fe, bb)
    same(got, ref, "slice_dev")
    assert int(got[3]["n_sync"]) == len(ref[1])
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    dib, res = fe.run_dev(t)
    nd = int(parse_results(res)[0]["n_dibits"])
    assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])
    for _ in range(3):
        dib, res = fe.run_dev_pipelined(t, dibits=dib, result=res)
    fe.join_dev()
    nd = int(parse_results(res)[0]["n_dibits"])
    assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])
    # the streaming calls cannot run 3.8c: with P25FE_CLOCK_CAUSAL_OK they give mode 1's answer ...
    fe3 = FE(symbol_clock=2 | 0x100)
    got3 = np.concatenate([fe3.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)])
    assert np.array_equal(got3, one[0])
    dib, res = fe3.run_dev(t)                                        # (the resident calls of such a handle still run 3.8c)
    nd = int(parse_results(res)[0]["n_dibits"])
    assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])


def test_mode2_is_refused_where_it_cannot_run(O, FE):
    """ABI 6: a handle created with symbol_clock = 2 asked for SPEC 3.8c.  The calls that see the stream in pieces -- host
    streaming chunks, host windows, the passes of a time shard (and so p25fe_shard_create's step) -- have 3.8b's causal rule only: they
    return P25FE_ERR_ARG instead of another receiver's dibits, move no state, and the resident calls of the same handle keep working.
    With P25FE_CLOCK_CAUSAL_OK or-ed in they run, as mode 1."""
    import torch
    from p25rx_amd import c4fm, _lib
    from p25rx_amd.frontend import parse_results
    iq = c4fm.synth(1.0, seed=5, snr_db=25.0, clock_ppm=100.0)[0]
    iq = iq[:len(iq) // 8 * 8]
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    bb = O.Demod().feed_cf32(iq)
    fe = FE(symbol_clock=2)
    refused = 0
    for call in (lambda: fe.run_cf32(iq[:16384]), lambda: fe.run_u8(c4fm.to_u8(iq[:16384])), lambda: fe.slice(bb[:3000]),
                 lambda: fe.run_host_windows(iq, window=65536), lambda: fe.shard_pass1(t, 0, len(iq), 0)):
        with pytest.raises(_lib.P25feError) as e:
            call()
        refused += e.value.status == _lib.ERR_ARG
    assert refused == 5
    dib, res = fe.run_dev(t)                                         # nothing moved: a fresh-stream resident call gives 3.8c's dibits
    ref = O.recv_range(bb, O.make_config(symbol_clock=2))
    nd = int(parse_results(res)[0]["n_dibits"])
    assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0])
    ok = FE(symbol_clock=2 | _lib.CLOCK_CAUSAL_OK)
    one = oracle_recv(O, bb, 1)
    assert np.array_equal(np.concatenate([ok.run_cf32(iq[o:o + 16384]) for o in range(0, len(iq), 16384)]), one[0])
    got, _ = FE(symbol_clock=2 | _lib.CLOCK_CAUSAL_OK).run_host_windows(iq, window=65536)
    assert np.array_equal(got, one[0])


def test_reslice_dense_events_and_lock_drops(O, FE):
    """Mode 2 where the list of detections is long and ragged: dozens of detections per tile, lock drops at decision indices, between
    detections, many in one tile -- lock runs of one detection (nothing to look ahead to), of two, intervals that fail 3.8b's test."""
    from p25rx_amd import c4fm
    rng = np.random.default_rng(79)
    pieces = []
    for fd, snr, amp, toff, ppm in ((26, 25.0, 0.5, 0, 0.0), (140, 14.0, 0.25, 3, 200.0), (31, 30.0, 0.1, 41, -90.0), (1200, 25.0, 0.4, 7, 250.0)):
        pieces.append(c4fm.synth(0.6, seed=400 + fd, snr_db=snr, frame_dibits=fd, amplitude=amp, timing_offset=toff, clock_ppm=ppm)[0])
        pieces.append((0.3 * (rng.standard_normal(5003) + 1j * rng.standard_normal(5003))).astype(np.complex64))
    bb = O.Demod().feed_cf32(np.concatenate(pieces))
    cfg = O.make_config(symbol_clock=2)
    ref0 = O.recv_range(bb, cfg)
    assert len(ref0[1]) > 200
    same(dev_slice(FE(symbol_clock=2), bb, sync_cap=8192), ref0, "no drops")
    sp = ref0[1]
    drops = [int(sp[10]) + 7, int(sp[10]) + 8, int(sp[11]) + 6, int(sp[40]) + 9, int(sp[41]) + 10, int(sp[41]) + 10, int(sp[90]) + 100,
             0, 1, len(bb) - 1, len(bb) + 7] + [int(x) for x in rng.integers(0, len(bb), size=80)]
    ref = O.recv_range(bb, cfg, drops)
    got = dev_slice(FE(symbol_clock=2), bb, resync=drops, sync_cap=8192)
    same(got, ref, "91 drops")
    assert np.array_equal(oracle_recv(O, bb, 1, drops)[1], ref[1])            # the same sync words as the streaming rule finds


def test_tracking_clock_on_a_nominal_clock_equals_the_fixed_stride(O, FE, c4fm_1s):
    bb = O.Demod().feed_cf32(c4fm_1s[0])
    fix, trk = dev_slice(FE(), bb), dev_slice(FE(symbol_clock=1), bb)
    n = len(trk[0])
    assert len(fix[0]) - 1 <= n <= len(fix[0]) and np.array_equal(trk[0], fix[0][:n])
    assert np.array_equal(trk[1], fix[1]) and np.array_equal(trk[2], fix[2])


@pytest.mark.parametrize("mode", [0, 1])
def test_dense_events_general_receiver(O, FE, mode):
    """Dozens of detections per tile, with and without lock drops between them, under both clocks."""
    from p25rx_amd import c4fm
    rng = np.random.default_rng(78)
    pieces = []
    for fd, snr, amp, toff in ((26, 25.0, 0.5, 0), (140, 14.0, 0.25, 3), (31, 30.0, 0.1, 41), (1200, 25.0, 0.4, 7)):
        pieces.append(c4fm.synth(0.6, seed=300 + fd, snr_db=snr, frame_dibits=fd, amplitude=amp, timing_offset=toff)[0])
        pieces.append((0.3 * (rng.standard_normal(5003) + 1j * rng.standard_normal(5003))).astype(np.complex64))
    bb = O.Demod().feed_cf32(np.concatenate(pieces))
    ref0 = oracle_recv(O, bb, mode)
    assert len(ref0[1]) > 200
    same(dev_slice(FE(symbol_clock=mode), bb, resync=[len(bb) + 1000]), ref0, "no drops")     # (a drop past the end: the general kernels run in mode 0 too)
    # lock drops: right at decision indices of dense detections, between them, many in one tile, duplicates
    sp = ref0[1]
    drops = [int(sp[10]) + 5, int(sp[10]) + 6, int(sp[11]) + 4, int(sp[40]) + 5 + 2 * mode, int(sp[41]) + 6 + 2 * mode,
             int(sp[41]) + 6 + 2 * mode, int(sp[90]) + 100] + [int(x) for x in rng.integers(0, len(bb), size=60)]
    ref = oracle_recv(O, bb, mode, drops)
    same(dev_slice(FE(symbol_clock=mode), bb, resync=drops), ref, "60 drops")


@pytest.mark.parametrize("mode", [0, 1])
def test_resync_inside_device_ranges(O, FE, mode):
    """MessageReceiver::resync at sample indices inside ONE device range (src/recv.rs:127-137, 179): at a sync word's
    decision index, one sample either side, inside an event-free tile, on tile boundaries, at 0, past the end."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    iq, truth, _ = c4fm.synth(2.5, seed=5, snr_db=28.0)
    iq = iq[:len(iq) // 8 * 8]
    bb = O.Demod().feed_cf32(iq)
    L = 2 * mode
    free = oracle_recv(O, bb, mode)
    sp = [int(x) for x in free[1]]
    assert len(sp) >= 13
    e = [s + 5 for s in sp]                                          # decision indices (s + W)
    tile = 7680
    quiet = [t for t in range(len(bb) // tile) if not any(t * tile <= x + L < (t + 1) * tile for x in e)]
    assert quiet
    cases = {
        "at the decision index": [e[2] + L], "one before": [e[2] + L - 1], "one after": [e[2] + L + 1], "two after": [e[2] + L + 2],
        "event-free tile": [quiet[0] * tile + 3000], "tile boundary": [2 * tile], "tile boundary + 1": [2 * tile + 1],
        "tile boundary - 1": [3 * tile - 1], "at zero": [0], "at one": [1], "last sample": [len(bb) - 1], "past the end": [len(bb) + 50],
        "inside a sync word": [sp[4] - 100], "several": [e[1] + L, e[3] + L + 1, quiet[0] * tile + 10, quiet[0] * tile + 11, e[7] + L - 1, 9 * tile],
    }
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    for name, drops in cases.items():
        ref = oracle_recv(O, bb, mode, drops)
        fe = FE(symbol_clock=mode)
        same(dev_slice(fe, bb, resync=drops), ref, name)
        if name in ("event-free tile", "several", "inside a sync word", "tile boundary"):
            assert len(ref[0]) < len(free[0])                        # symbols between the drop and the next sync word are lost
        # the list is consumed by the call: the next one on the handle runs free
        same(dev_slice(fe, bb), free, name + " (next call)")
        # fused path
        fe.resync_at_dev(torch.tensor(sorted(drops), dtype=torch.int64, device="cuda"))
        dib, res = fe.run_dev(t)
        nd = int(parse_results(res)[0]["n_dibits"])
        assert nd == len(ref[0]) and np.array_equal(dib[0, :nd].cpu().numpy(), ref[0]), name + " (run_dev)"


@pytest.mark.parametrize("mode", [0, 1])
def test_lock_drops_through_the_streaming_calls(O, FE, mode):
    """p25fe_resync_at_dev before a host-buffer call (p25fe_slice, p25fe_run_cf32): the chunk is the range, the indices stay
    ABSOLUTE, the list is consumed by that call -- also when the chunk would otherwise take the one-launch path."""
    import torch
    from p25rx_amd import c4fm
    iq, _, _ = c4fm.synth(1.5, seed=9, snr_db=28.0)
    iq = iq[:len(iq) // 16384 * 16384]
    bb = O.Demod().feed_cf32(iq)
    L = 2 * mode
    free = oracle_recv(O, bb, mode)
    sp = [int(x) + 5 + L for x in free[1]]
    drops = [sp[1], sp[3] + 1, sp[3] + 700, sp[5] - 1]
    ref = oracle_recv(O, bb, mode, drops)
    assert len(ref[0]) < len(free[0])

    def feed(fe, call, sizes, n_total, units):
        """chunks of the given sizes; before each chunk the drops that fall inside it go to the handle"""
        parts, o = [], 0
        while o < n_total:
            n = min(sizes[len(parts) % len(sizes)], n_total - o)
            lo, hi = units(o), units(o + n)
            mine = [q for q in drops if lo <= q < hi]
            if mine:
                fe.resync_at_dev(torch.tensor(mine, dtype=torch.int64, device="cuda"))
            parts.append(call(o, n))
            o += n
        return parts

    fe = FE(symbol_clock=mode)
    parts = feed(fe, lambda o, n: fe.slice(bb[o:o + n]), [3277, 3276, 9000, 1, 2500], len(bb), lambda o: o)
    same([np.concatenate([p[k] for p in parts]) for k in range(3)], ref, "p25fe_slice")
    fe = FE(symbol_clock=mode)
    n_bb = lambda o: (o - 5) // 5 + 1 if o > 4 else 0                  # baseband samples produced by the first o IQ samples (p25fe_n_baseband)
    parts = feed(fe, lambda o, n: fe.run_cf32(iq[o:o + n]), [16384, 16384, 49152], len(iq), n_bb)
    got = np.concatenate(parts)
    assert len(got) == len(ref[0]) and np.array_equal(got, ref[0])


@pytest.mark.parametrize("mode", [0, 1])
def test_time_shards_with_tracking_clock_and_lock_drops(O, FE, mode):
    """Config 5's shard / resolve / pass 2 with the general receiver: carry-in clocks across shards, a shard whose only
    event is a lock drop, a shard with one detection that takes its period from the previous shard."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results, n_baseband
    iq, _, _ = c4fm.synth(3.0, seed=34, snr_db=24.0, frame_dibits=1500, clock_ppm=80.0 if mode else 0.0)
    iq = iq[:len(iq) // 8 * 8]
    bb = O.Demod().feed_cf32(iq)
    free = oracle_recv(O, bb, mode)
    sp = [int(x) for x in free[1]]
    drops = [sp[3] + 5 + 2 * mode, 100000 // 5 + 4000, 310006 // 5 + 50, 310006 // 5 + 60]
    t = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda()
    # ([0, 1]: lock drops in front of the stream's first samples -- under the tracking clock's lookahead they sit at -2 / -1,
    # and the summary's carry_end must still say "this shard has an event")
    for rs in ([], drops, [0, 1] + drops, [0]):
        ref = oracle_recv(O, bb, mode, rs)
        fe = FE(symbol_clock=mode)
        halo = fe.shard_halo()
        cuts = [0, 100004, 100004 + 3002, 200008, 310006, 400000, len(iq)]
        fes = [FE(symbol_clock=mode) for _ in range(len(cuts) - 1)]
        d_rs = torch.tensor(sorted(rs), dtype=torch.int64, device="cuda") if rs else None
        summ, bb0, bbn = [], [], []
        for r in range(len(cuts) - 1):
            a, b = cuts[r], cuts[r + 1]
            h = min(a, halo)
            fes[r].resync_at_dev(d_rs)
            if r % 2:
                fes[r].shard_pass1_main(t[a - h:b], offset=h, n_hist=h, abs0=a)
                res = fes[r].shard_pass1_finish(t[a - h:b], offset=h, n_hist=h, abs0=a)
            else:
                res = fes[r].shard_pass1(t[a - h:b], offset=h, n_hist=h, abs0=a)
            summ.append(parse_results(res)[0])
            bb0.append(n_baseband(0, a))
            bbn.append(n_baseband(a, b - a))
        anc, off = fe.shard_resolve(np.array(summ), bb0, bbn)
        out = []
        for r in range(len(cuts) - 1):
            dib, res = fes[r].shard_pass2(anc[r:r + 1], bbn[r], t.device)
            k = int(parse_results(res)[0]["n_dibits"])
            assert off[r] == sum(len(x) for x in out), "shard %d offset" % r
            out.append(dib[0, :k].cpu().numpy())
        got = np.concatenate(out)
        assert int(off[-1]) == len(got) == len(ref[0]) and np.array_equal(got, ref[0]), "drops %s" % rs
        summ_t = torch.from_numpy(np.frombuffer(np.array(summ).tobytes(), dtype=np.uint8).copy()).view(len(summ), -1).cuda()
        d_anc, d_off = fe.shard_resolve_dev(summ_t, torch.tensor(bb0, dtype=torch.int64, device="cuda"),
                                            torch.tensor(bbn, dtype=torch.int64, device="cuda"))
        assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), off) and d_anc.cpu().numpy().tobytes() == anc.tobytes()


def test_dibit_row_capacity_is_a_hard_bound(O, FE, c4fm_1s):
    """dibit_stride is the capacity: a row too short for the range's dibits is filled to the brim, the count stays exact,
    nothing is written past the row (a clock offset makes a range hold more than n / 10 dibits)."""
    import ctypes as C
    import torch
    from p25rx_amd._lib import RESULT_DTYPE
    from p25rx_amd.frontend import parse_results
    bb = O.Demod().feed_cf32(c4fm_1s[0])
    ref = O.Recv().feed(bb)[0]
    fe = FE()
    t = torch.from_numpy(bb).cuda()
    for cap in (0, 1, 777, len(ref) - 1, len(ref), len(ref) + 5):
        buf = torch.full((cap + 4096,), 0xEE, dtype=torch.uint8, device="cuda")
        res = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        fe._chk(fe.L.p25fe_slice_dev(fe.h, C.c_void_p(t.data_ptr()), len(bb), 0, len(bb), 0, None,
                                     C.c_void_p(buf.data_ptr() + 2048), cap, None, None, 0, C.c_void_p(res.data_ptr()),
                                     fe._stream()))
        assert int(parse_results(res)[0]["n_dibits"]) == len(ref)
        got = buf.cpu().numpy()
        k = min(cap, len(ref))
        assert np.array_equal(got[2048:2048 + k], ref[:k])
        assert np.all(got[:2048] == 0xEE) and np.all(got[2048 + k:] == 0xEE), "cap %d" % cap


def test_mode2_pipelined_three_deep_while_the_detection_list_grows(O, FE):
    """the slicer by detection keeps ONE list of detections per handle (evrec / evnext / evoff), shared by all pipelined lanes --
    safe because the receive kernels of consecutive calls run in order on one stream, and a list that has to GROW is freed and
    re-allocated (hipFree synchronises the device).  Calls of growing size, three scratch sets deep, enqueued back to back: every call's
    dibits are the oracle's two-pass answer for ITS capture."""
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    os.environ["P25FE_PIPE_DEPTH"] = "3"
    try:
        fe = FE(symbol_clock=2)
    finally:
        os.environ.pop("P25FE_PIPE_DEPTH", None)
    caps, refs = [], []
    for k, (secs, fd, ppm) in enumerate(((0.4, 200, 120.0), (0.9, 150, -180.0), (1.7, 96, 200.0), (0.5, 864, 0.0), (2.6, 64, 150.0))):
        iq = c4fm.synth(secs, seed=90 + k, snr_db=28.0, frame_dibits=fd, clock_ppm=ppm)[0]
        iq = iq[:len(iq) // 8 * 8]
        caps.append(torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).cuda())
        refs.append(O.recv_range(O.Demod().feed_cf32(iq), O.make_config(symbol_clock=2))[0])
    outs = []
    for rnd in range(2):                                             # second round: the list no longer grows, the lanes still rotate
        for t in caps:
            outs.append(fe.run_dev_pipelined(t))                     # (fresh outputs per call: nothing is read before the join)
    fe.join_dev()
    torch.cuda.synchronize()
    for k, (dib, res) in enumerate(outs):
        ref = refs[k % len(caps)]
        nd = int(parse_results(res)[0]["n_dibits"])
        assert nd == len(ref) and np.array_equal(dib[0, :nd].cpu().numpy(), ref), k


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", [1, 0])
def test_general_receiver_over_more_than_64_groups(O, FE, mode):
    """The general receiver's hierarchical scan walks the groups of a channel 64 at a time (k_scan_tiles_g's last workgroup: one group
    per lane, the carried state handed from step to step).  700 s of one channel are 4 375 tiles = 69 groups -- two steps --, with
    lock drops in groups of both steps, at a step's first group, and stretches of event-free groups (whole groups of noise) that force
    the sequential walk beside the parallel common case.  Tracking clock (mode 1) and the fixed stride with a lock-drop list (mode 0):
    dibits, sync positions and sync dibit indices equal the oracle's; the frames between clean sync words are the modulator's."""
    import torch
    from p25rx_amd import c4fm
    n = 700 * 240000
    dev = torch.device("cuda", 0)
    iq_t, truth = c4fm.synth_torch(n, seed=1234, device=dev, snr_db=30.0, clock_ppm=100.0 if mode else 0.0)
    # two silent stretches (no sync words for > one group of 64 tiles = 491 520 baseband samples): noise only
    g = torch.Generator(device=dev); g.manual_seed(7)
    # (the first one 42 s long: the sync word behind it takes its period from an interval of 2 M samples, and the slicer's per-instant
    # arithmetic for that clock no longer fits 32 bits -- the tiles of that frame take the 64-bit form beside everyone else's narrow one)
    for a0, a1 in ((40 * 240000, 82 * 240000), (655 * 240000, 669 * 240000)):
        iq_t[a0:a1] = 0.02 * torch.randn((a1 - a0, 2), generator=g, device=dev)
    iq = iq_t.cpu().numpy().view(np.complex64).reshape(-1)
    bb = O.Demod().feed_cf32(iq)
    tile, group = 7680, 64 * 7680
    drops = [3 * group + 100, 3 * group + 101, 64 * group, 64 * group + 5 * tile + 17, 66 * group - 1, 20 * group + 3 * tile, len(bb) - 5]
    ref = oracle_recv(O, bb, mode, drops)
    assert len(ref[1]) > 3000
    fe = FE(symbol_clock=mode)
    got = dev_slice(fe, bb, resync=drops, sync_cap=8192)
    same(got, ref, "700 s, %d lock drops" % len(drops))
    r = got[3]
    assert int(r["n_sync"]) == len(ref[1]) and int(r["n_dibits"]) == len(ref[0])


def test_slicer_instant_arithmetic_narrow_and_wide_agree(FE):
    """k_slice_g places instant j of a clock (s, D, N) at s + (j D) div N with phase ((j D) mod N) 64 div N: per call one 64-bit division
    of j_lo D, then 32-bit arithmetic per instant whenever the numbers fit, the 64-bit form otherwise.  The clocks D / N and k D / k N
    are the same clock; with k = 1 700 nothing fits 32 bits any more.  A range without a sync word, sliced under a carry-in anchor with
    either clock: the same dibits, and as many as the integers say."""
    import torch
    from p25rx_amd.frontend import parse_results
    n_bb = 3 * 7680 + 1234
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    t = 0.3 * torch.randn(n_bb + 64, generator=g, device="cuda")
    fe = FE(symbol_clock=1)
    outs = []
    for k in (1, 1700):
        D, N = (4 * 8641 + 1) * k, 4 * 864 * k
        s0 = -40
        dib, res, _, _ = fe.slice_dev(t, n_bb, anchor_in=[(s0, 0.2, 0.0, -0.2, 1, D, N)])
        torch.cuda.synchronize()
        r = parse_results(res)[0]
        assert int(r["n_sync"]) == 0, "noise held a sync word: pick another seed"
        # instants j >= 1 at s0 + (j D) div N inside the range (the tracking clock decides the last few samples one call later)
        want = sum(1 for j in range(1, n_bb // 10 + 64) if max(0, s0 + 6) <= s0 + (j * D) // N < n_bb)
        nd = int(r["n_dibits"])
        assert want - 1 <= nd <= want, (k, nd, want)
        outs.append(dib[0, :nd].cpu().numpy())
    assert len(outs[0]) == len(outs[1]) and np.array_equal(outs[0], outs[1])
    assert len(set(outs[0].tolist())) == 4                           # (a real slicer output, not a row of zeros)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_more_than_64_detections_in_one_tile(O, FE, mode):
    """The slicers keep their tile's detections in registers, 64 at a time (one per lane, read by v_readlane), and reload for a tile
    that holds more; K2 remembers the thresholds of 64.  Real traffic never gets there (a sync word is 240 samples long), but the
    ten polyphase planes are independent sample sets: every plane carries its own back-to-back sync words, plane r's ending at
    positions 240 m + 21 r -- ten detections per 240 samples, 21 or 51 samples apart, 320 per tile.  Dibits, sync positions and
    sync dibit indices equal the oracle's under all three clocks."""
    mask = 0x050cdf                                                  # P25FE_SYNC_SIGN_MASK: bit j = sync symbol j (oldest first) is +3
    pat = np.array([1.0 if (mask >> j) & 1 else -1.0 for j in range(24)], dtype=np.float32)
    n_bb = 3 * 7680 + 777
    q = np.arange(n_bb, dtype=np.int64)
    r = q % 10
    bb = (0.24 * pat[((q - 21 * r + 230) // 10) % 24]).astype(np.float32)
    bb += (0.004 * np.random.default_rng(11).standard_normal(n_bb)).astype(np.float32)
    ref = oracle_recv(O, bb, mode)
    per_tile = np.bincount((ref[1] + 5) // 7680, minlength=3)        # (decision index = sync position + W)
    assert per_tile.max() > 256, per_tile
    same(dev_slice(FE(symbol_clock=mode), bb, sync_cap=4096), ref, "320 detections per tile, mode %d" % mode)
    # ... and with lock drops between them (fixed stride and causal clock: the lock-drop list makes both take the general kernels)
    if mode < 2:
        drops = [int(ref[1][70]) + 7, int(ref[1][200]) + 6, int(ref[1][201]) + 9, 2 * 7680 + 3]
        same(dev_slice(FE(symbol_clock=mode), bb, resync=drops, sync_cap=4096), oracle_recv(O, bb, mode, drops), "with drops, mode %d" % mode)
    # ... and through the streaming chunks of the reference's size (one wave runs detection, scan and slicer of a chunk: k_recv_chunk;
    # 3 276 samples hold 136 of these detections)
    if mode == 0:
        fe2 = FE()
        got_d, got_s = [], []
        for o in range(0, n_bb, 3276):
            dd, ss, _ = fe2.slice(bb[o:o + 3276])
            got_d.append(dd); got_s.append(ss)
        assert np.array_equal(np.concatenate(got_d), ref[0]) and np.array_equal(np.concatenate(got_s), ref[1])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_golden_dense_planes_fixture(FE, mode):
    """The committed vector of the densest scene (tests/golden/dense_planes.npz, written by gen_golden_dense.py from the oracle and
    checked against the independent model on the CPU side): the HIP receiver reproduces it from the file alone -- no generator, no
    oracle in between -- for the three clocks, without and with lock drops."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_planes.npz"))
    bb = g["bb_bits"].view(np.float32)
    for tag, drops in (("", None), ("_drops", [int(x) for x in g["drops"]])):
        want = (g["dibits_m%d%s" % (mode, tag)], g["sync_pos_m%d%s" % (mode, tag)], g["sync_dibit_m%d%s" % (mode, tag)].astype(np.uint64))
        same(dev_slice(FE(symbol_clock=mode), bb, resync=drops, sync_cap=2048), want, "golden dense planes, mode %d%s" % (mode, tag))
