#!/usr/bin/env python3
"""Randomised parity run: HIP path (through the C ABI) against the CPU oracle on seeded random scenes -- test
infrastructure, like everything that imports `oracle`.

    python tests/fuzz_parity.py [--minutes M | --cases N] [--seed S]

A scene draws: 1 - 3 channels, capture length, SNR (down to where false and missed sync words happen), frame length (down to
back-to-back sync words), carrier offset, sample-clock error, amplitude, input format (cf32 / u8), NaN / Inf / huge samples,
tap tables (the build's, or random ones of random length up to the ABI's 64), symbol clock (fixed / tracking), lock drops at
random indices (device lists per channel, and MessageReceiver::resync between chunks), a random chunking for the streaming
entry points with the handle's state exported and imported at a random boundary.  Compared bit for bit: baseband (linear,
device and host forms; power within 1e-3 dB), dibits, sync positions, sync dibit indices -- for the device-resident forms
(demod_dev, slice_dev, run_dev, run_dev_pipelined), the streaming forms (demod_*, slice, run_*) under that chunking, and
time shards at random cut points (one- and two-launch pass 1, lock drops, host and device resolve).  Any difference
prints the scene's seed and stops with exit code 1; `tests/test_gpu_fuzz.py` runs a few fixed seeds under pytest."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def random_taps(rng, n):
    """A low-pass-ish random table: a windowed sinc with random cutoff plus noise, unity DC gain (float32)."""
    k = np.arange(n) - (n - 1) / 2.0
    fc = rng.uniform(0.05, 0.3)
    h = np.sinc(2 * fc * k) * np.hanning(n + 2)[1:-1] + rng.normal(0, 0.01, n)
    h = h / h.sum()
    return [float(np.float32(v)) for v in h]


def scene(seed):
    rng = np.random.default_rng(seed)
    s = {"seed": seed}
    s["seconds"] = float(rng.choice([0.05, 0.2, 0.5, 1.0, 2.0], p=[0.1, 0.25, 0.3, 0.25, 0.1]))
    s["snr_db"] = float(rng.choice([30.0, 20.0, 12.0, 8.0, 5.0, 2.0]))
    s["frame"] = int(rng.choice([24, 30, 48, 100, 300, 864, 864, 3000]))
    s["freq"] = float(rng.choice([0.0, 0.0, 150.0, -400.0, 900.0]))
    s["ppm"] = float(rng.choice([0.0, 0.0, 0.0, 40.0, -100.0, 250.0]))
    s["amp"] = float(rng.choice([0.5, 0.05, 0.9]))
    s["fmt"] = str(rng.choice(["cf32", "cf32", "u8"]))
    s["clock"] = int(rng.integers(0, 3))                          # 0 fixed stride, 1 tracking, 2 tracking + SPEC 3.8c in the resident calls
    s["taps"] = int(rng.choice([0, 0, 0, 1, 2]))             # 0: the build's, 1: random <= 31 / 41, 2: random up to 64 / 64
    s["n_drops"] = int(rng.choice([0, 0, 1, 3, 20]))
    s["timing"] = int(rng.integers(0, 50))
    s["channels"] = int(rng.choice([1, 1, 1, 2, 3]))
    if s["seconds"] <= 0.2 and rng.integers(0, 8) == 0:
        s["channels"] = int(rng.choice([17, 64]))            # a real batch now and then (short captures): channel indexing at scale
    s["damage"] = int(rng.choice([0, 0, 0, 0, 1]))           # cf32 only: NaN / Inf / huge / denormal samples sprinkled in
    # ragged lengths: around multiples of a receiver tile (7 680 baseband = 38 400 IQ samples), of a K1 sub-tile (1 600) and anything
    s["extra"] = int(rng.choice([0, 8, 1600, 1608, 38400 - 8, 38400, 38400 + 8, 2 * 38400, int(rng.integers(0, 50000)) // 8 * 8]))
    s["cut"] = int(rng.choice([0, 0, 1]))                    # 1: the capture is cut down to (a multiple of 38 400) + extra
    # ABI 4: the run-time arguments of the reference's constructors (drawn LAST: the scenes of earlier rounds keep their numbers)
    s["fm"] = int(rng.choice([0, 0, 0, 1, 2]))               # 0: 5 kHz / 48 kHz, 1: another deviation, 2: an explicit scale
    s["u8tab"] = int(rng.choice([0, 0, 0, 1, 2, 3]))         # 0: fma(b, 2/255, -1), 1: (b - 127) / 128 as scale / offset,
    #                                                          2: correctly rounded (b - 127.5) / 127.5 as a TABLE (not affine), 3: a random monotone table
    s["kernels"] = int(rng.choice([0, 0, 0, 0, 0, 1]))       # 1: kernels specialised for the scene's numbers (hipRTC: seconds per scene)
    # ABI 5 (drawn after everything above, for the same reason): MovingAverage::new(10) as a table, Decimator::new(5)'s phase
    s["avg"] = int(rng.choice([0, 0, 0, 0, 1, 2, 3]))        # 0: ten taps of 0.1, 1: another moving average, 2: random <= 11 taps, 3: random 12 .. 64 taps
    s["phase"] = int(rng.choice([4, 4, 4, 0, 1, 2, 3]))
    return s, rng


def run_scene(seed, O, FE, torch, verbose=False):
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results
    s, rng = scene(seed)

    def n_baseband(a, n, ph=s["phase"]):                            # docs/SPEC.md 3.2 with the scene's decimator phase
        o0 = (ph + 5 - a % 5) % 5
        return (n - o0 - 1) // 5 + 1 if n > o0 else 0
    Cn = s["channels"]
    n_iq = int(round(s["seconds"] * 240000)) // 8 * 8
    if s["cut"]:
        n_iq = max(n_iq // 38400, 0) * 38400
    n_iq = max(n_iq + s["extra"], 4000)
    lead = 4
    nsym = n_iq // 50 - 2 * lead
    iqs = []
    for c in range(Cn):
        d = c4fm.make_dibits(nsym, seed + 7919 * c, s["frame"])
        x, _ = c4fm.modulate(d, snr_db=s["snr_db"], seed=seed + 7919 * c, freq_offset_hz=s["freq"], amplitude=s["amp"],
                             timing_offset=(s["timing"] + 13 * c) % 50, lead_symbols=lead, clock_ppm=s["ppm"])
        iqs.append(np.ascontiguousarray(x[:n_iq]))
    n_iq = min(len(x) for x in iqs) // 8 * 8
    iq = np.stack([x[:n_iq] for x in iqs])                            # [C, n] complex64
    if s["damage"] and s["fmt"] == "cf32" and n_iq > 4000:
        f = iq.view(np.float32)
        vals = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-42, -0.0, 3.0e38], dtype=np.float32)
        for k in range(12):
            f[int(rng.integers(0, Cn)), int(rng.integers(0, 2 * n_iq))] = vals[k % len(vals)]
    dt = ct = None
    if s["taps"] == 1:
        dt, ct = random_taps(rng, int(rng.integers(1, 32))), random_taps(rng, int(rng.integers(1, 42)))
    elif s["taps"] == 2:
        dt, ct = random_taps(rng, int(rng.integers(32, 65))), random_taps(rng, int(rng.integers(42, 65)))
    xkw = {}
    if s["fm"] == 1:
        xkw["fm_deviation_hz"] = int(rng.choice([2500, 4000, 6250]))
    elif s["fm"] == 2:
        xkw["fm_gain"] = float(np.float32(rng.uniform(0.5, 3.0)))
    if s["u8tab"] == 1:
        xkw["u8_scale"], xkw["u8_offset"] = float(np.float32(1 / 128.0)), float(np.float32(-127 / 128.0))
    elif s["u8tab"] == 2:
        xkw["u8_lut"] = ((np.arange(256, dtype=np.float64) - 127.5) / 127.5).astype(np.float32)
    elif s["u8tab"] == 3:
        xkw["u8_lut"] = (np.cumsum(rng.uniform(0.0, 0.016, 256)) - 1.0).astype(np.float32)
    if s["avg"] == 1:
        L = int(rng.integers(1, 33))
        xkw["avg_taps"] = [float(np.float32(1.0 / L))] * L
    elif s["avg"] == 2:
        xkw["avg_taps"] = random_taps(rng, int(rng.integers(2, 12)))
    elif s["avg"] == 3:
        xkw["avg_taps"] = random_taps(rng, int(rng.integers(12, 65)))
    if s["phase"] != 4:
        xkw["decim_phase"] = s["phase"]
    cfg = O.make_config(decim_taps=dt, chan_taps=ct, symbol_clock=s["clock"], **xkw)
    fkw = dict(xkw)
    if s["kernels"] == 1:
        fkw["specialize"] = 2                                        # FORCE: the build's own numbers go through hipRTC too
    # (mode 2 with P25FE_CLOCK_CAUSAL_OK: the scene runs the streaming / window / shard forms too, which have the causal rule only)
    mk = lambda C_=Cn: FE(n_channels=C_, decim_taps=dt, chan_taps=ct, symbol_clock=(0x102 if s["clock"] == 2 else s["clock"]), **fkw)
    what = []

    def check(name, ok):
        what.append(name)
        if not ok:
            raise AssertionError("seed %d: %s differs; scene %r" % (seed, name, s))

    def same_f(a, b):          # floats: bit for bit where the oracle has a number, NaN where it has a NaN
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        na, nb_ = np.isnan(a), np.isnan(b)
        return a.shape == b.shape and np.array_equal(na, nb_) and np.array_equal(bits(a[~na]), bits(b[~nb_]))

    rows = lambda x: [x] if Cn == 1 else list(x)                    # the wrappers return a bare array for one channel

    # ---- oracle, per channel
    raw = np.stack([c4fm.to_u8(iq[c]) for c in range(Cn)]) if s["fmt"] == "u8" else iq
    ref_bb, ref_pw = [], []
    for c in range(Cn):
        dm = O.Demod(cfg)
        b, pw = (dm.feed_u8(raw[c], want_power=True) if s["fmt"] == "u8" else dm.feed_cf32(raw[c], want_power=True))
        ref_bb.append(b); ref_pw.append(pw)
    nb = len(ref_bb[0])
    # lock drops: random indices (every channel its own list) + some boundaries of the slice pass's chunking (host resync())
    plan, po = [], 0
    while po < nb:
        n = min(int(rng.choice([1, 3, 239, 241, 3276, 3277, 9000, int(rng.integers(1, 30000))])), nb - po)
        plan.append((po, n)); po += n
    host_rs = sorted(set(int(plan[int(k)][0]) for k in rng.integers(0, len(plan), size=int(rng.choice([0, 0, 1, 3]))))) if plan else []
    drops = [sorted(set([int(x) for x in rng.integers(0, nb + 5, size=s["n_drops"])] + host_rs)) for c in range(Cn)]
    n_list = max(len(x) for x in drops)

    def oracle_recv(c):
        r = O.Recv(cfg)
        outs, o = [], 0
        for q in drops[c]:
            q = min(max(q, 0), nb)
            outs.append(r.feed(ref_bb[c][o:q]))
            r.resync()
            o = q
        outs.append(r.feed(ref_bb[c][o:]))
        return (np.concatenate([x[0] for x in outs]), np.concatenate([x[1] for x in outs]),
                np.concatenate([x[2] for x in outs]).astype(np.uint64))
    ref = [oracle_recv(c) for c in range(Cn)]
    # the calls that hold the whole range follow SPEC 3.8c in mode 2 (the oracle's two passes); every other form keeps the causal rule
    res_ref = ref if s["clock"] != 2 else [tuple(x.astype(np.uint64) if k == 2 else x for k, x in enumerate(O.recv_range(ref_bb[c], cfg, drops[c])))
                                           for c in range(Cn)]

    def set_drops(fe, lo=None, hi=None, skip=()):
        """the drops inside [lo, hi) (all of them: None) as a device list [C, n], short rows padded with INT64_MAX"""
        mine = [[q for q in drops[c] if (lo is None or lo <= q < hi) and q not in skip] for c in range(Cn)]
        k = max(len(x) for x in mine)
        if k:
            arr = np.full((Cn, k), np.iinfo(np.int64).max, dtype=np.int64)
            for c in range(Cn):
                arr[c, :len(mine[c])] = mine[c]
            fe.resync_at_dev(torch.from_numpy(arr).cuda())

    # ---- device-resident forms
    if s["fmt"] == "u8":
        t = torch.from_numpy(raw.reshape(Cn, -1, 2)).cuda()
    else:
        t = torch.from_numpy(np.ascontiguousarray(raw).view(np.float32).reshape(Cn, -1, 2)).cuda()
    fe = mk()
    bb, nb_g, pw = fe.demod_dev(t, want_power=True)
    check("demod_dev length", nb_g == nb)
    for c in range(Cn):
        check("demod_dev baseband", same_f(bb[c, :nb].cpu().numpy(), ref_bb[c]))
        p_, q_ = float(pw[c].item()), float(ref_pw[c])
        check("demod_dev power", (np.isnan(p_) and np.isnan(q_)) or p_ == q_ or abs(p_ - q_) <= 1e-3 or not np.isfinite(q_))
    set_drops(fe)
    scap = nb // 6 + 2
    dib, res, sp, sd = fe.slice_dev(bb[:, :nb].contiguous(), nb, sync_cap=scap)
    rr = parse_results(res)
    for c in range(Cn):
        nd, ns = int(rr[c]["n_dibits"]), int(rr[c]["n_sync"])
        check("slice_dev counts", nd == len(res_ref[c][0]) and ns == len(res_ref[c][1]))
        k = min(nd, dib.shape[1])
        check("slice_dev dibits", np.array_equal(dib[c, :k].cpu().numpy(), res_ref[c][0][:k]))
        check("slice_dev sync_pos", np.array_equal(sp[c, :ns].cpu().numpy(), res_ref[c][1]))
        check("slice_dev sync_dibit", np.array_equal(sd[c, :ns].cpu().numpy().astype(np.uint64), res_ref[c][2]))
    for name in ("run_dev", "run_dev_pipelined", "run_dev_pipelined"):
        set_drops(fe)
        dib2, res2 = getattr(fe, name)(t)
        if name == "run_dev_pipelined":
            fe.join_dev()
        torch.cuda.synchronize()
        r2 = parse_results(res2)
        for c in range(Cn):
            nd2 = int(r2[c]["n_dibits"])
            k = min(nd2, dib2.shape[1])
            check(name, nd2 == len(res_ref[c][0]) and np.array_equal(dib2[c, :k].cpu().numpy(), res_ref[c][0][:k]))

    # ---- mixed sequence on ONE handle without joining in between: two pipelined calls (their receive kernels run on the handle's
    # stream), then a call that overwrites the same scratch -- the library has to order them itself
    set_drops(fe)
    dA, rA = fe.run_dev_pipelined(t)
    dB, rB = fe.run_dev_pipelined(t)
    set_drops(fe)
    dib3, res3, sp3, sd3 = fe.slice_dev(bb[:, :nb].contiguous(), nb, sync_cap=scap)
    dC, rC = fe.run_dev(t)
    fe.join_dev()
    torch.cuda.synchronize()
    rA_, rB_, r3_, rC_ = parse_results(rA), parse_results(rB), parse_results(res3), parse_results(rC)
    free = None
    for c in range(Cn):
        ndA, ndB, nd3, ndC = int(rA_[c]["n_dibits"]), int(rB_[c]["n_dibits"]), int(r3_[c]["n_dibits"]), int(rC_[c]["n_dibits"])
        check("mixed: pipelined A", ndA == len(res_ref[c][0]) and np.array_equal(dA[c, :min(ndA, dA.shape[1])].cpu().numpy(), res_ref[c][0][:dA.shape[1]]))
        check("mixed: slice_dev", nd3 == len(res_ref[c][0]) and np.array_equal(dib3[c, :min(nd3, dib3.shape[1])].cpu().numpy(), res_ref[c][0][:dib3.shape[1]]))
        # B and C ran without a drop list (the list is consumed by the call it was set for): the free-running receiver
        if not any(drops[c_] for c_ in range(Cn)):
            check("mixed: pipelined B", ndB == len(res_ref[c][0]) and np.array_equal(dB[c, :min(ndB, dB.shape[1])].cpu().numpy(), res_ref[c][0][:dB.shape[1]]))
            check("mixed: run_dev C", ndC == len(res_ref[c][0]) and np.array_equal(dC[c, :min(ndC, dC.shape[1])].cpu().numpy(), res_ref[c][0][:dC.shape[1]]))
        else:
            check("mixed: B == C", ndB == ndC and np.array_equal(dB[c, :min(ndB, dB.shape[1])].cpu().numpy(), dC[c, :min(ndC, dC.shape[1])].cpu().numpy()))

    # ---- streaming forms under a random chunking (IQ samples per call; u8: two bytes per sample); the handle's state goes
    # through export / import at one random boundary
    ch, o = [], 0
    while o < n_iq:
        n = min(int(rng.choice([1, 2, 7, 333, 16384, 16384, 40000, int(rng.integers(1, 100000))])), n_iq - o)
        ch.append((o, n)); o += n
    swap_at = int(rng.integers(0, len(ch))) if ch else -1
    fe_d, fe_r = mk(), mk()
    parts_bb, parts_d = [[] for _ in range(Cn)], [[] for _ in range(Cn)]
    for k, (o, n) in enumerate(ch):
        if k == swap_at:
            blob = fe_r.state_export()
            fe_r = mk()
            fe_r.state_import(blob)
        if s["fmt"] == "u8":
            pb = fe_d.demod_u8(raw[:, 2 * o:2 * (o + n)])
        else:
            pb = fe_d.demod_cf32(raw[:, o:o + n])
        set_drops(fe_r, n_baseband(0, o), n_baseband(0, o + n))
        pd = fe_r.run_u8(raw[:, 2 * o:2 * (o + n)]) if s["fmt"] == "u8" else fe_r.run_cf32(raw[:, o:o + n])
        for c in range(Cn):
            parts_bb[c].append(rows(pb)[c]); parts_d[c].append(rows(pd)[c])
    for c in range(Cn):
        check("streaming baseband", same_f(np.concatenate(parts_bb[c]), ref_bb[c]))
        gd = np.concatenate(parts_d[c])
        check("streaming run", len(gd) == len(ref[c][0]) and np.array_equal(gd, ref[c][0]))
    fe_s = mk()
    parts = [[] for _ in range(Cn)]
    for po, n in plan:
        if po in host_rs:
            fe_s.resync()                                            # MessageReceiver::resync between two chunks (src/recv.rs:136)
        set_drops(fe_s, po, po + n, skip=[po] if po in host_rs else ())
        out = fe_s.slice(np.stack([ref_bb[c][po:po + n] for c in range(Cn)]))
        for c in range(Cn):
            parts[c].append(out if Cn == 1 else out[c])
    for c in range(Cn):
        if parts[c]:
            cat = [np.concatenate([p_[k] for p_ in parts[c]]) for k in range(3)]
            check("streaming slice", all(len(cat[k]) == len(ref[c][k]) and np.array_equal(cat[k], ref[c][k].astype(cat[k].dtype)) for k in range(3)))

    # ---- time shards at random (8-aligned) cut points: one channel, every shard handed the whole drop list, the two-launch
    # form on odd shards, host and device resolve
    if n_iq >= 64 and Cn == 1:
        k = int(rng.integers(2, 6))
        cuts = sorted(set([0, n_iq] + [int(x) // 8 * 8 for x in rng.integers(8, n_iq, size=k - 1)]))
        t1 = t[0]
        fe1 = mk(1)
        halo = fe1.shard_halo()
        d_rs = torch.tensor(drops[0], dtype=torch.int64, device="cuda") if drops[0] else None
        fes, summ, bb0, bbn = [], [], [], []
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            h = min(a, halo)
            f = mk(1)
            f.resync_at_dev(d_rs)
            if i % 2:
                f.shard_pass1_main(t1[a - h:b], offset=h, n_hist=h, abs0=a)
                rs = f.shard_pass1_finish(t1[a - h:b], offset=h, n_hist=h, abs0=a)
            else:
                rs = f.shard_pass1(t1[a - h:b], offset=h, n_hist=h, abs0=a)
            summ.append(parse_results(rs)[0])
            bb0.append(n_baseband(0, a)); bbn.append(n_baseband(a, b - a)); fes.append(f)
        anc_h, off_h = fe1.shard_resolve(np.array(summ), bb0, bbn)
        summ_t = torch.from_numpy(np.frombuffer(np.array(summ).tobytes(), dtype=np.uint8).copy()).view(len(fes), -1).cuda()
        anc, off = fe1.shard_resolve_dev(summ_t, torch.tensor(bb0, dtype=torch.int64, device="cuda"),
                                         torch.tensor(bbn, dtype=torch.int64, device="cuda"))
        offs = off.cpu().numpy()
        check("shard resolve host == device", np.array_equal(offs.astype(np.uint64), off_h) and anc.cpu().numpy().tobytes() == anc_h.tobytes())
        got_rows = []
        for i, f in enumerate(fes):
            dd, rs2 = f.shard_pass2(anc[i:i + 1], bbn[i], t.device)
            kk = int(parse_results(rs2)[0]["n_dibits"])
            check("shard %d count" % i, kk == int(offs[i + 1] - offs[i]))
            got_rows.append(dd[0, :kk].cpu().numpy())
        got_all = np.concatenate(got_rows) if got_rows else np.zeros(0, np.uint8)
        if not np.array_equal(got_all, ref[0][0]):
            m = min(len(got_all), len(ref[0][0]))
            s["detail"] = {"cuts": cuts, "bb0": bb0, "offsets": [int(x) for x in offs], "len": (len(got_all), len(ref[0][0])),
                           "first_diff": [int(x) for x in np.nonzero(got_all[:m] != ref[0][0][:m])[0][:4]], "drops": drops[0],
                           "syncs": [int(x) for x in ref[0][1][:12]], "sync_dibit": [int(x) for x in ref[0][2][:12]]}
        check("time shards", np.array_equal(got_all, ref[0][0]))
    if verbose:
        print("seed %d ok: %r -> %d baseband, %s dibits, %s syncs, %d chunks" % (seed, s, nb, [len(r_[0]) for r_ in ref], [len(r_[1]) for r_ in ref], len(ch)))
    return len(what)


def dense_scene(seed):
    """A baseband no transmitter makes: the ten polyphase planes are independent sample sets, each active one carries its own
    back-to-back sync words at its own alignment -- up to 320 detections per 7 680-sample tile (the slicers hold 64 in registers at a
    time, K2 remembers the thresholds of 64), some of them closer than the peak test allows; any clock, lock drops.
    -> (symbol_clock, baseband float32, lock-drop indices).  Its own generator: a scene is reproducible on the CPU from its seed."""
    rng = np.random.default_rng(seed ^ 0xd15e)
    mode = int(rng.choice([0, 0, 1, 2]))
    pat = np.array([1.0 if (0x050cdf >> j) & 1 else -1.0 for j in range(24)], dtype=np.float32)
    n_bb = int(rng.choice([3000, 7680, 7681, 2 * 7680 + 11, int(rng.integers(5000, 4 * 7680))]))
    q = np.arange(n_bb, dtype=np.int64)
    active = rng.random(10) < float(rng.choice([0.3, 0.7, 1.0]))
    align = np.where(rng.random(10) < 0.5, 21 * np.arange(10), 10 * rng.integers(0, 24, size=10) + np.arange(10))
    amp = float(rng.choice([0.1, 0.24, 0.5]))
    bb = (amp * pat[((q - align[q % 10] + 230) // 10) % 24]).astype(np.float32)
    bb = np.where(active[q % 10], bb, 0.0).astype(np.float32)
    bb += (float(rng.choice([0.002, 0.02, 0.08])) * rng.standard_normal(n_bb)).astype(np.float32)
    drops = sorted(set(int(x) for x in rng.integers(1, n_bb, size=int(rng.choice([0, 0, 2, 6])))))
    return mode, bb, drops


def run_aux_scene(seed, O, FE, torch, verbose=False):
    """The forms around the fresh-stream ones: an owned range INSIDE a longer device buffer (random start, random amount of
    history, absolute decimation grid), the wideband stages K0 (bit-exact) and K6 (fp64 oracle, 2e-6 tolerance) on random
    lengths / offsets, multi-channel K0, and network identifiers with random bit errors."""
    from p25rx_amd import c4fm
    from p25rx_amd._lib import NID_DTYPE
    from p25rx_amd.frontend import n_baseband
    rng = np.random.default_rng(seed ^ 0x5a5a5a)
    what = []

    def check(name, ok, **info):
        what.append(name)
        if not ok:
            raise AssertionError("aux seed %d: %s differs; %r" % (seed, name, info))

    # ---- (a) range inside a buffer: demod_dev(offset, n_hist, abs0), cf32 and u8
    n_iq = int(rng.choice([4000, 24000, 120000])) // 8 * 8
    iq, _, _ = c4fm.synth(n_iq / 240000.0 + 0.01, seed=seed, snr_db=float(rng.choice([30.0, 8.0])), frame_dibits=int(rng.choice([100, 864])))
    iq = np.ascontiguousarray(iq[:n_iq])
    fmt = str(rng.choice(["cf32", "u8"]))
    raw = c4fm.to_u8(iq) if fmt == "u8" else iq
    t = torch.from_numpy(raw.reshape(-1, 2) if fmt == "u8" else raw.view(np.float32).reshape(-1, 2)).cuda()
    fe = FE()
    for _ in range(4):
        a = int(rng.integers(0, n_iq // 8)) * 8                      # owned range [a, b): 16-byte aligned start for both formats
        b = int(rng.integers(a, n_iq + 1))
        h = int(min(a, rng.choice([0, 8, 40, 280, 284, 288, 1000, 2048])))
        if fmt == "u8":
            # (zeros in front of a short history cannot be written as bytes -- 127.5 is no byte: the u8 form is compared where the
            # history is complete or reaches the start of the stream)
            ref_all = O.Demod().feed_u8(raw) if (a - h == 0 or h >= 284) else None
        else:
            z = np.array(raw, copy=True)
            z[:a - h] = 0                                            # what lies in front of the history does not exist: zeros, as at
            ref_all = O.Demod().feed_cf32(z)                         # the start of a stream
        if ref_all is None:
            continue
        p0, p1 = n_baseband(0, a), n_baseband(0, b)
        bb, nb = fe.demod_dev(t[:b], n_hist=h, abs0=a, offset=a)
        check("demod_dev range length", nb == p1 - p0, a=a, b=b, h=h, fmt=fmt)
        check("demod_dev range", np.array_equal(bits(bb[0, :nb].cpu().numpy()), bits(ref_all[p0:p1])), a=a, b=b, h=h, fmt=fmt)

    # ---- (b) K0 on 1 - 3 channels of random wideband samples, whole and as a range with history; K6 on the first channel
    Cw = int(rng.choice([1, 1, 2, 3]))
    nw = int(rng.choice([1000, 20000, 100000])) // 2 * 2 + int(rng.choice([0, 2, 6]))
    wide = (rng.normal(0, 0.3, (Cw, nw)) + 1j * rng.normal(0, 0.3, (Cw, nw))).astype(np.complex64)
    tw = torch.from_numpy(wide.view(np.float32).reshape(Cw, nw, 2)).cuda()
    few = FE(n_channels=Cw)
    refs = [O.PreDecim().feed(wide[c]) for c in range(Cw)]
    y, no = few.predecim_dev(tw)
    check("predecim length", no == len(refs[0]), nw=nw)
    for c in range(Cw):
        check("predecim", np.array_equal(bits(y[c, :no].cpu().numpy().reshape(-1)), bits(refs[c].view(np.float32))), nw=nw, c=c)
    if Cw == 1 and nw > 400:
        off = int(rng.integers(1, nw // 2)) * 2
        hh = int(min(off, rng.choice([79, 80, 96, 500])))
        y2, no2 = few.predecim_dev(tw[0], n_hist=hh, abs0=off, offset=off)
        first = off // 10                                            # outputs whose instant 10 m + 9 lies in front of the range
        check("predecim range", np.array_equal(bits(y2[0, :no2].cpu().numpy().reshape(-1)),
                                               bits(refs[0][first:first + no2].view(np.float32))), nw=nw, off=off, hh=hh)
        if nw <= 20000:
            spec = O.load_spec()
            refc = O.channelise(wide[0])
            yc, nc = few.channelise_dev(tw[0])
            tol = 2e-6 * float(np.sum(np.abs(np.array(spec["pre_taps"], dtype=np.float64)))) * float(np.abs(wide[0]).max())
            got = yc[:, :nc].cpu().numpy().view(np.complex64)[:, :, 0]
            check("channelise", nc == refc.shape[1] and float(np.abs(got - refc).max()) <= tol, nw=nw)

    # ---- (c) network identifiers: frames with a valid NID, random bit errors in the dibit stream
    nf = int(rng.integers(3, 12))
    nidf = lambda f: (int((seed * 31 + f * 7) & 0xfff), int((f * 5 + seed) & 0xf))
    d = c4fm.make_dibits(nf * 150 + 60, seed, 150, nid=nidf)
    flips = rng.integers(0, len(d), size=int(rng.choice([0, 5, 40, 200])))
    d = np.array(d, copy=True)
    d[flips] ^= rng.integers(1, 4, size=len(flips)).astype(np.uint8)
    sdib = np.array([24 + 150 * f for f in range(nf)], dtype=np.uint64)
    spos = np.array([2400 + 1500 * f for f in range(nf)], dtype=np.int64)
    refn = O.nid_decode(d, sdib, spos)
    out = np.zeros(nf, dtype=NID_DTYPE)
    import ctypes as C_
    fe._chk(fe.L.p25fe_nid(fe.h, d.ctypes.data_as(C_.c_void_p), len(d), sdib.ctypes.data_as(C_.c_void_p), spos.ctypes.data_as(C_.c_void_p),
                           nf, out.ctypes.data_as(C_.c_void_p)))
    check("nid", out.tobytes() == refn.tobytes(), nf=nf, flips=len(flips))
    # ---- (d) the receiver on a baseband no transmitter makes (dense_scene below)
    from p25rx_amd.frontend import parse_results
    mode, bb, drops = dense_scene(seed)
    n_bb = len(bb)
    if mode == 2:
        refd = O.recv_range(bb, O.make_config(symbol_clock=2), drops)
    else:
        rcv = O.Recv(O.make_config(symbol_clock=mode))
        outs, o = [], 0
        for c in drops + [n_bb]:
            outs.append(rcv.feed(bb[o:c]))
            if c < n_bb:
                rcv.resync()
            o = c
        refd = [np.concatenate([x[k] for x in outs]) for k in range(3)]
    fed = FE(symbol_clock=mode)
    if drops:
        fed.resync_at_dev(torch.tensor(drops, dtype=torch.int64, device="cuda"))
    dib, res, sp, sd = fed.slice_dev(torch.from_numpy(bb).cuda(), n_bb, sync_cap=2048)
    rr = parse_results(res)[0]
    nd, ns = int(rr["n_dibits"]), int(rr["n_sync"])
    check("dense planes: counts", nd == len(refd[0]) and ns == len(refd[1]), mode=mode, n_bb=n_bb, nd=nd, ns=ns, want=(len(refd[0]), len(refd[1])))
    kd = min(nd, dib.shape[1])                                       # (the row's capacity bounds the stores, the count stays exact)
    got_d = dib[0, :kd].cpu().numpy()
    bad = np.nonzero(got_d != refd[0][:kd])[0]
    check("dense planes: dibits", len(bad) == 0, mode=mode, n_bb=n_bb, nd=nd, cap=int(dib.shape[1]), drops=drops,
          first_bad=[int(x) for x in bad[:8]], n_bad=len(bad))
    check("dense planes: sync", np.array_equal(sp[0, :ns].cpu().numpy(), refd[1])
          and np.array_equal(sd[0, :ns].cpu().numpy().astype(np.uint64), np.asarray(refd[2]).astype(np.uint64)), mode=mode, n_bb=n_bb, ns=ns)
    fed.close()
    if verbose:
        print("aux seed %d ok: %d comparisons" % (seed, len(what)))
    return len(what)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=None)
    ap.add_argument("--cases", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    import torch
    from oracle import oracle as O
    from p25rx_amd.frontend import FrontEnd as FE
    t0, n, checks = time.time(), 0, 0
    while True:
        if a.cases is not None and n >= a.cases:
            break
        if a.minutes is not None and (time.time() - t0) > a.minutes * 60:
            break
        if a.cases is None and a.minutes is None and n >= 20:
            break
        try:
            checks += run_scene(a.seed + n, O, FE, torch, a.v)
            if n % 4 == 0:
                checks += run_aux_scene(a.seed + n, O, FE, torch, a.v)
        except AssertionError as e:
            print("FAIL", e)
            sys.exit(1)
        n += 1
    print("fuzz_parity: %d scenes (seeds %d..%d), %d comparisons, all bit-exact, %.0f s" % (n, a.seed, a.seed + n - 1, checks, time.time() - t0))


if __name__ == "__main__":
    main()
