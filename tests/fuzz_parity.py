#!/usr/bin/env python3
"""Randomised parity run: HIP path (through the C ABI) against the CPU oracle on seeded random scenes -- test
infrastructure, like everything that imports `oracle`.

    python tests/fuzz_parity.py [--minutes M | --cases N] [--seed S]

A scene draws: capture length, SNR (down to where false and missed sync words happen), frame length (down to back-to-back
sync words), carrier offset, sample-clock error, amplitude, input format (cf32 / u8), tap tables (the build's, or random
ones of random length up to the ABI's 64), symbol clock (fixed / tracking), lock drops at random indices, and a random
chunking for the streaming entry points.  Compared bit for bit: baseband (linear, device and host forms), dibits, sync
positions, sync dibit indices -- for the device-resident forms (demod_dev, slice_dev, run_dev, run_dev_pipelined), the
streaming forms (demod_*, slice, run_*) under that chunking, and time shards at random cut points.  Any difference
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
    s["clock"] = int(rng.integers(0, 2))
    s["taps"] = int(rng.choice([0, 0, 0, 1, 2]))             # 0: the build's, 1: random <= 31 / 41, 2: random up to 64 / 64
    s["n_drops"] = int(rng.choice([0, 0, 1, 3, 20]))
    s["timing"] = int(rng.integers(0, 50))
    return s, rng


def run_scene(seed, O, FE, torch, verbose=False):
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import parse_results, n_baseband
    s, rng = scene(seed)
    n_iq = int(round(s["seconds"] * 240000)) // 8 * 8
    lead = 4
    nsym = n_iq // 50 - 2 * lead
    d = c4fm.make_dibits(nsym, seed, s["frame"])
    iq, _ = c4fm.modulate(d, snr_db=s["snr_db"], seed=seed, freq_offset_hz=s["freq"], amplitude=s["amp"], timing_offset=s["timing"],
                          lead_symbols=lead, clock_ppm=s["ppm"])
    iq = np.ascontiguousarray(iq[:n_iq])
    n_iq = len(iq) // 8 * 8
    iq = iq[:n_iq]
    dt = ct = None
    if s["taps"] == 1:
        dt, ct = random_taps(rng, int(rng.integers(1, 32))), random_taps(rng, int(rng.integers(1, 42)))
    elif s["taps"] == 2:
        dt, ct = random_taps(rng, int(rng.integers(32, 65))), random_taps(rng, int(rng.integers(42, 65)))
    cfg = O.make_config(decim_taps=dt, chan_taps=ct, symbol_clock=s["clock"])
    mk = lambda **kw: FE(decim_taps=dt, chan_taps=ct, symbol_clock=s["clock"], **kw)
    what = []

    def check(name, ok):
        what.append(name)
        if not ok:
            raise AssertionError("seed %d: %s differs; scene %r" % (seed, name, s))

    # ---- oracle
    if s["fmt"] == "u8":
        raw = c4fm.to_u8(iq)
        ref_bb = O.Demod(cfg).feed_u8(raw)
    else:
        raw = iq
        ref_bb = O.Demod(cfg).feed_cf32(iq)
    nb = len(ref_bb)
    drops = sorted(int(x) for x in rng.integers(0, nb + 5, size=s["n_drops"])) if s["n_drops"] else []
    r = O.Recv(cfg)
    outs, o = [], 0
    for q in drops:
        q = min(max(q, 0), nb)
        outs.append(r.feed(ref_bb[o:q]))
        r.resync()
        o = q
    outs.append(r.feed(ref_bb[o:]))
    ref = (np.concatenate([x[0] for x in outs]), np.concatenate([x[1] for x in outs]), np.concatenate([x[2] for x in outs]).astype(np.uint64))

    def set_drops(fe, lo=None, hi=None):
        mine = [q for q in drops if (lo is None or lo <= q < hi)]
        if mine:
            fe.resync_at_dev(torch.tensor(mine, dtype=torch.int64, device="cuda"))

    # ---- device-resident forms
    if s["fmt"] == "u8":
        t = torch.from_numpy(raw.reshape(-1, 2)).cuda()
    else:
        t = torch.from_numpy(raw.view(np.float32).reshape(-1, 2)).cuda()
    fe = mk()
    bb, nb_g = fe.demod_dev(t)
    check("demod_dev length", nb_g == nb)
    g = bb[0, :nb].cpu().numpy()
    check("demod_dev baseband", np.array_equal(bits(g), bits(ref_bb)))
    set_drops(fe)
    dib, res, sp, sd = fe.slice_dev(bb[:, :nb].contiguous(), nb, sync_cap=nb // 6 + 2)
    rr = parse_results(res)[0]
    nd, ns = int(rr["n_dibits"]), int(rr["n_sync"])
    check("slice_dev counts", nd == len(ref[0]) and ns == len(ref[1]))
    check("slice_dev dibits", np.array_equal(dib[0, :nd].cpu().numpy(), ref[0]))
    check("slice_dev sync_pos", np.array_equal(sp[0, :ns].cpu().numpy(), ref[1]))
    check("slice_dev sync_dibit", np.array_equal(sd[0, :ns].cpu().numpy().astype(np.uint64), ref[2]))
    for name in ("run_dev", "run_dev_pipelined"):
        set_drops(fe)
        dib2, res2 = getattr(fe, name)(t)
        if name == "run_dev_pipelined":
            fe.join_dev()
        torch.cuda.synchronize()
        nd2 = int(parse_results(res2)[0]["n_dibits"])
        check(name, nd2 == len(ref[0]) and np.array_equal(dib2[0, :nd2].cpu().numpy(), ref[0]))

    # ---- streaming forms under a random chunking (IQ samples per call; u8: two bytes per sample)
    def chunks(total):
        o, out = 0, []
        while o < total:
            n = int(rng.choice([1, 2, 7, 333, 16384, 16384, 40000, int(rng.integers(1, 100000))]))
            n = min(n, total - o)
            out.append((o, n))
            o += n
        return out

    ch = chunks(n_iq)
    fe_d, fe_r = mk(), mk()
    parts_bb, parts_d = [], []
    for o, n in ch:
        if s["fmt"] == "u8":
            parts_bb.append(fe_d.demod_u8(raw[2 * o:2 * (o + n)]))
        else:
            parts_bb.append(fe_d.demod_cf32(raw[o:o + n]))
        set_drops(fe_r, n_baseband(0, o), n_baseband(0, o + n))
        parts_d.append(fe_r.run_u8(raw[2 * o:2 * (o + n)]) if s["fmt"] == "u8" else fe_r.run_cf32(raw[o:o + n]))
    gbb = np.concatenate(parts_bb) if parts_bb else np.zeros(0, np.float32)
    check("streaming baseband", len(gbb) == nb and np.array_equal(bits(gbb), bits(ref_bb)))
    gd = np.concatenate(parts_d) if parts_d else np.zeros(0, np.uint8)
    check("streaming run", len(gd) == len(ref[0]) and np.array_equal(gd, ref[0]))
    fe_s = mk()
    po, parts = 0, []
    while po < nb:
        n = min(int(rng.choice([1, 3, 239, 241, 3276, 3277, 9000, int(rng.integers(1, 30000))])), nb - po)
        set_drops(fe_s, po, po + n)
        parts.append(fe_s.slice(ref_bb[po:po + n]))
        po += n
    if parts:
        cat = [np.concatenate([p[k] for p in parts]) for k in range(3)]
        check("streaming slice", all(len(cat[k]) == len(ref[k]) and np.array_equal(cat[k], ref[k].astype(cat[k].dtype)) for k in range(3)))

    # ---- time shards at random (8-aligned) cut points, cf32 and u8 alike
    if n_iq >= 64 and not drops:
        k = int(rng.integers(2, 5))
        cuts = sorted(set([0, n_iq] + [int(x) // 8 * 8 for x in rng.integers(8, n_iq, size=k - 1)]))
        halo = fe.shard_halo()
        fes, summ, bb0, bbn = [], [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            h = min(a, halo)
            f = mk()
            rs = f.shard_pass1(t[a - h:b], offset=h, n_hist=h, abs0=a)
            summ.append(parse_results(rs)[0])
            bb0.append(n_baseband(0, a)); bbn.append(n_baseband(a, b - a)); fes.append(f)
        summ_t = torch.from_numpy(np.frombuffer(np.array(summ).tobytes(), dtype=np.uint8).copy()).view(len(fes), -1).cuda()
        anc, off = fe.shard_resolve_dev(summ_t, torch.tensor(bb0, dtype=torch.int64, device="cuda"),
                                        torch.tensor(bbn, dtype=torch.int64, device="cuda"))
        offs = off.cpu().numpy()
        rows = []
        for i, f in enumerate(fes):
            dd, rs2 = f.shard_pass2(anc[i:i + 1], bbn[i], t.device)
            kk = int(parse_results(rs2)[0]["n_dibits"])
            check("shard %d count" % i, kk == int(offs[i + 1] - offs[i]))
            rows.append(dd[0, :kk].cpu().numpy())
        check("time shards", np.array_equal(np.concatenate(rows) if rows else np.zeros(0, np.uint8), ref[0]))
    if verbose:
        print("seed %d ok: %r -> %d baseband, %d dibits, %d syncs, %d chunks" % (seed, s, nb, len(ref[0]), len(ref[1]), len(ch)))
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
        except AssertionError as e:
            print("FAIL", e)
            sys.exit(1)
        n += 1
    print("fuzz_parity: %d scenes (seeds %d..%d), %d comparisons, all bit-exact, %.0f s" % (n, a.seed, a.seed + n - 1, checks, time.time() - t0))


if __name__ == "__main__":
    main()
