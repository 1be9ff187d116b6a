"""Seeded synthetic C4FM source (stands in for the RTL-SDR reader of src/sdr.rs:25-33).

The reference has no fake SDR backend and no recorded fixtures, so the
tests and bench.py drive the hot path with a C4FM modulator whose symbols are known:
demod(mod(d)) == d is the end-to-end truth statement.

Modulation follows the public TIA-102.BAAA description: 4800 symbols/s, dibits
01/00/10/11 -> +3/+1/-1/-3, deviation +-1800/+-600 Hz, transmit shaping = raised cosine
(alpha = 0.2) x inverse-sinc P(f); the receiver's integrate-and-dump D(f) is the 10-sample
moving average of src/demod.rs:52,114.  Frames: 24-dibit frame-sync word 0x5575F5FF77FF
followed by `frame_dibits - 24` PRNG dibits.

Two back ends with the same structure: numpy (CPU, fixtures and small tests) and torch
(on-device generation of BASELINE.json's 600 s inputs; torch is plumbing here).
"""
import numpy as np

FS_IQ = 240000
FS_BB = 48000
BAUD = 4800
SPS_IQ = FS_IQ // BAUD            # 50
SYNC_WORD = 0x5575F5FF77FF
SYNC_DIBITS = 24
DIBIT_TO_SYMBOL = np.array([1, 3, -1, -3], dtype=np.float64)   # index = dibit value 00,01,10,11
DEV_HZ_PER_UNIT = 600.0           # +-3 -> +-1800 Hz


def sync_dibits():
    return np.array([(SYNC_WORD >> (2 * (SYNC_DIBITS - 1 - i))) & 3 for i in range(SYNC_DIBITS)], dtype=np.uint8)


def tx_filter(span_symbols=8, sps=SPS_IQ, alpha=0.2):
    """Impulse response of RC(alpha) x P(f) sampled at `sps` per symbol, by frequency sampling.

    Normalised so that (h * one-symbol boxcar)(0) == 1: after the receiver's integrate-and-
    dump, an isolated symbol of value a reads a at its centre.
    """
    n = 2 * span_symbols * sps + 1
    nfft = 1 << int(np.ceil(np.log2(n * 8)))
    f = np.fft.fftfreq(nfft, d=1.0 / (sps * BAUD))          # Hz
    af = np.abs(f)
    f1, f2 = (1 - alpha) * BAUD / 2, (1 + alpha) * BAUD / 2
    rc = np.where(af <= f1, 1.0, np.where(af <= f2, 0.5 * (1 + np.cos(np.pi * (af - f1) / (f2 - f1))), 0.0))
    x = np.pi * f / BAUD
    p = np.where(af <= f2, np.where(np.abs(x) < 1e-12, 1.0, x / np.sin(np.where(np.abs(x) < 1e-12, 1.0, x))), 0.0)
    h = np.real(np.fft.ifft(rc * p))
    h = np.roll(h, n // 2)[:n]
    h *= np.hanning(n + 2)[1:-1] ** 0.25                     # gentle taper of the truncation
    box = np.ones(sps) / sps
    g = np.convolve(h, box)
    h /= g[(len(g) - 1) // 2] * sps / sps
    return h


NID_GEN_POLY = int("6331141367235453", 8)      # BCH(63,16,23), TIA-102.BAAA (verified in tools/gen_spec.py)


def nid_word(nac, duid):
    """64-bit NID: systematic BCH(63,16) code word of (nac << 4 | duid) followed by one extra (even parity) bit."""
    m = ((nac & 0xFFF) << 4 | (duid & 0xF)) << 47
    r, db = m, NID_GEN_POLY.bit_length()
    while r.bit_length() >= db:
        r ^= NID_GEN_POLY << (r.bit_length() - db)
    cw = m | r
    return (cw << 1) | (bin(cw).count("1") & 1)


def nid_dibits(nac, duid, status=2):
    """The 33 dibits that follow a frame sync: 32 NID dibits with the status symbol at position 11."""
    w = nid_word(nac, duid)
    d = [(w >> (62 - 2 * i)) & 3 for i in range(32)]
    return np.array(d[:11] + [status] + d[11:], dtype=np.uint8)


def make_dibits(n_dibits, seed, frame_dibits=864, nid=None):
    """PRNG dibits with the frame-sync word every `frame_dibits`; nid = callable(frame_index) -> (nac, duid) puts a
    valid network identifier (with its status symbol) after every sync word."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = rng.integers(0, 4, size=n_dibits, dtype=np.uint8)
    fs = sync_dibits()
    for f, start in enumerate(range(0, n_dibits - SYNC_DIBITS + 1, frame_dibits)):
        d[start:start + SYNC_DIBITS] = fs
        if nid is not None and start + SYNC_DIBITS + 33 <= n_dibits:
            nac, duid = nid(f)
            d[start + SYNC_DIBITS:start + SYNC_DIBITS + 33] = nid_dibits(nac, duid)
    return d


def modulate(dibits, snr_db=30.0, seed=0, freq_offset_hz=0.0, amplitude=0.5, timing_offset=0, lead_symbols=4,
             clock_ppm=0.0):
    """dibits -> cf32 IQ at 240 ksps.  Returns (iq, info).

    info['symbol_center_iq'][k] = IQ sample index of the centre of symbol k (before any
    receiver delay).  `timing_offset` shifts the whole waveform by that many IQ samples.
    `clock_ppm`: the receiver's sample clock runs that many ppm FAST relative to the transmitter's symbol clock (a
    symbol then lasts 50 (1 + ppm 1e-6) received samples): the phase trajectory is resampled before the noise is added.
    """
    dibits = np.asarray(dibits, dtype=np.uint8)
    sym = DIBIT_TO_SYMBOL[dibits]
    h = tx_filter()
    nsym = len(sym)
    up = np.zeros((nsym + 2 * lead_symbols) * SPS_IQ, dtype=np.float64)
    centers = (np.arange(nsym) + lead_symbols) * SPS_IQ + SPS_IQ // 2 + timing_offset
    # impulse train weighted so that the integrate-and-dump of the shaped train reads `sym`
    idx = centers
    np.add.at(up, idx, sym * SPS_IQ)
    shaped = np.convolve(up, h, mode="same")
    # the boxcar in the receiver integrates frequency, so `shaped` is the frequency in units
    freq = shaped * DEV_HZ_PER_UNIT / SPS_IQ * 1.0 + freq_offset_hz
    phase = 2.0 * np.pi * np.cumsum(freq) / FS_IQ
    if clock_ppm:
        k = 1.0 + clock_ppm * 1e-6
        t = np.arange(int(len(phase) * k)) / k                   # receiver sample m looks at transmitter time m / k
        phase = np.interp(t, np.arange(len(phase)), phase)
        centers = centers * k
    iq = amplitude * np.exp(1j * phase)
    if snr_db is not None:
        rng = np.random.Generator(np.random.PCG64(seed + 0x5EED))
        sigma = amplitude * 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0)
        iq = iq + sigma * (rng.standard_normal(len(iq)) + 1j * rng.standard_normal(len(iq)))
    return iq.astype(np.complex64), {"symbol_center_iq": centers, "n_symbols": nsym}


def synth(seconds=1.0, seed=1, snr_db=30.0, frame_dibits=864, nid=None, **kw):
    """`seconds` of C4FM IQ at 240 ksps (BASELINE.json config 1/2 shape).  Returns iq, dibits, info."""
    n_iq = int(round(seconds * FS_IQ))
    lead = kw.pop("lead_symbols", 4)
    nsym = n_iq // SPS_IQ - 2 * lead
    d = make_dibits(nsym, seed, frame_dibits, nid=nid)
    iq, info = modulate(d, snr_db=snr_db, seed=seed, lead_symbols=lead, **kw)
    return iq[:n_iq], d, info


def to_u8(iq):
    """cf32 -> interleaved RTL-SDR style u8 I/Q bytes (inverse of SPEC 3.1, rounded, clipped)."""
    z = np.empty(2 * len(iq), dtype=np.float64)
    z[0::2] = iq.real
    z[1::2] = iq.imag
    return np.clip(np.rint((z + 1.0) * 127.5), 0, 255).astype(np.uint8)


def align_dibits(decoded, truth, max_skip=64):
    """Find k, j such that decoded[k:] matches truth[j:], return (k, j, n_matched, n_errors).

    The receiver emits nothing before its first sync detection and then starts with the
    dibit after that sync word; this searches the first frame boundary.
    """
    decoded = np.asarray(decoded)
    truth = np.asarray(truth)
    best = None
    for j in range(0, min(len(truth), 4000)):
        n = min(len(decoded), len(truth) - j)
        if n < 64:
            break
        if np.array_equal(decoded[:48], truth[j:j + 48]):
            err = int(np.count_nonzero(decoded[:n] != truth[j:j + n]))
            best = (0, j, n, err)
            break
    return best


# ---------------------------------------------------------------------------------------------
# torch back end: generate long captures directly in HBM (bench.py, full-size tests)
# ---------------------------------------------------------------------------------------------
def synth_torch(n_iq, seed, device, snr_db=30.0, frame_dibits=864, amplitude=0.5, chunk_symbols=1 << 18,
                out=None, out_u8=None, clock_ppm=0.0):
    """Generate n_iq cf32 samples of C4FM on `device`.  Returns (iq[n_iq, 2] float32, dibits uint8 cpu).

    Generated in chunks of symbols with phase continuity carried in float64; each chunk is
    shaped by a polyphase matrix product over +-span symbols of context.
    clock_ppm: as in synth() -- the receiver's sample clock runs that many ppm fast: the phase trajectory is read at the instants
    m / (1 + ppm 1e-6) (linear interpolation of the float64 phase, whose curvature over one 240 kHz sample is < 1e-3 rad) before
    the noise is added.
    """
    import torch

    lead = 4
    nsym = n_iq // SPS_IQ - 2 * lead
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    d = torch.randint(0, 4, (nsym,), generator=g, dtype=torch.uint8)
    fs = torch.from_numpy(sync_dibits())
    starts = torch.arange(0, nsym - SYNC_DIBITS + 1, frame_dibits)
    idx = (starts[:, None] + torch.arange(SYNC_DIBITS)[None, :]).reshape(-1)
    d[idx] = fs.repeat(len(starts))
    sym_all = torch.from_numpy(DIBIT_TO_SYMBOL.astype(np.float32))[d.long()]

    h_np = tx_filter().astype(np.float32)
    centre = (len(h_np) - 1) // 2
    span = centre // SPS_IQ                                   # symbols of context each side
    # Pulse shaping as a polyphase product: the shaped sample at phase p of symbol s is sum_j sym[s + j] * hp[p][j] -- one
    # [symbols x (2 span + 1)] x [(2 span + 1) x 50] matrix product per chunk.  (It used to be a conv1d over the zero-stuffed
    # impulse train, which MIOpen ran as a naive convolution: 70 s of a 125 s bench run and 99 % of every kernel trace.)
    hp_np = np.zeros((2 * span + 1, SPS_IQ), dtype=np.float32)
    for j in range(-span, span + 1):
        for p_ in range(SPS_IQ):
            k = centre + p_ - SPS_IQ // 2 - SPS_IQ * j
            if 0 <= k < len(h_np):
                hp_np[j + span, p_] = h_np[k] * SPS_IQ
    hp = torch.from_numpy(hp_np).to(device)
    if out is None:
        out = torch.empty((n_iq, 2), dtype=torch.float32, device=device)
    phase0 = torch.zeros((), dtype=torch.float64, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 0x5EED)
    sigma = amplitude * 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0) if snr_db is not None else 0.0
    total_sym = nsym + 2 * lead
    padded = torch.zeros(total_sym + 2 * span, dtype=torch.float32)
    padded[span + lead: span + lead + nsym] = sym_all
    pos = 0
    kclk = 1.0 + float(clock_ppm) * 1e-6
    nom0 = 0                                                  # nominal index of the chunk's first sample
    carry = None                                              # the previous chunk's last phase sample (nominal index nom0 - 1)
    for s0 in range(0, total_sym, chunk_symbols):
        s1 = min(total_sym, s0 + chunk_symbols)
        seg = padded[s0: s1 + 2 * span].to(device)            # symbols s0-span .. s1+span
        y = (seg.unfold(0, 2 * span + 1, 1) @ hp).reshape(-1)    # [(s1 - s0) x 50] -> samples of symbols s0 .. s1
        freq = y.double() * (DEV_HZ_PER_UNIT / SPS_IQ)
        ph = torch.cumsum(freq, 0) * (2.0 * np.pi / FS_IQ) + phase0
        phase0 = torch.remainder(ph[-1], 2.0 * np.pi)
        if clock_ppm:
            ext = ph if carry is None else torch.cat([carry.reshape(1), ph])
            base = nom0 if carry is None else nom0 - 1
            m_hi = int(np.floor((nom0 + len(ph) - 1) * kclk))         # last output index whose instant lies inside this chunk
            nom0 += len(ph)
            carry = phase0.clone()                               # (the next chunk continues from the wrapped value)
            if m_hi < pos:
                continue
            tau = torch.arange(pos, m_hi + 1, dtype=torch.float64, device=device) / kclk - base
            i0 = torch.clamp(torch.floor(tau).long(), 0, len(ext) - 1)
            fr = tau - i0.double()
            i1 = torch.clamp(i0 + 1, max=len(ext) - 1)
            ph = ext[i0] * (1.0 - fr) + ext[i1] * fr
        ph = torch.remainder(ph, 2.0 * np.pi).float()
        n = min(len(ph), n_iq - pos)
        if n <= 0:
            break
        blk = out[pos:pos + n]
        blk[:, 0] = amplitude * torch.cos(ph[:n])
        blk[:, 1] = amplitude * torch.sin(ph[:n])
        if sigma:
            blk += sigma * torch.randn((n, 2), generator=gen, device=device, dtype=torch.float32)
        pos += n
    if pos < n_iq:
        out[pos:].zero_()
    return out, d.numpy()


# ---------------------------------------------------------------------------------------------
# wideband scene for the channeliser (SPEC 3.11): several C4FM carriers on the 12.5 kHz raster of a 2.4 Msps capture
# ---------------------------------------------------------------------------------------------
FS_WIDE = 2400000
CHZ_CHANNELS = 192


def synth_wideband(seconds, carriers, snr_db=25.0, seed=0, frame_dibits=864, nid=None):
    """carriers: {raster index c (0..191): (seed, relative amplitude)}.  Each carrier is c4fm.synth at 240 ksps,
    interpolated x10 and shifted to c * 12.5 kHz (c >= 96: negative frequencies).  Returns (wide complex64 @ 2.4 Msps,
    {c: truth dibits})."""
    from scipy import signal as sps
    n = int(round(seconds * FS_WIDE))
    wide = np.zeros(n, dtype=np.complex128)
    truth = {}
    k = np.arange(n)
    for c, (sd, amp) in carriers.items():
        iq, d, _ = synth(seconds, seed=sd, snr_db=None, frame_dibits=frame_dibits, nid=nid)
        up = sps.resample_poly(iq.astype(np.complex128), FS_WIDE // FS_IQ, 1)[:n]
        wide[:len(up)] += amp * up * np.exp(2j * np.pi * ((c * k[:len(up)]) % CHZ_CHANNELS) / CHZ_CHANNELS)
        truth[c] = d
    if snr_db is not None:
        rng = np.random.Generator(np.random.PCG64(seed + 0xC42))
        # noise density such that a 0.5-amplitude carrier has `snr_db` in ITS 240 kHz channel stream
        sigma = 0.5 * 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0) * np.sqrt(FS_WIDE / FS_IQ)
        wide += sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return wide.astype(np.complex64), truth
