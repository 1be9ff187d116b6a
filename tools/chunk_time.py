import time, numpy as np, sys
sys.path.insert(0, '.')
from p25rx_amd import c4fm
from p25rx_amd.frontend import FrontEnd
iq, truth, _ = c4fm.synth(2.0, seed=1, snr_db=30.0)
for fmt in ("cf32", "u8"):
    data = iq if fmt == "cf32" else c4fm.to_u8(iq)
    step = 16384 if fmt == "cf32" else 32768
    fe = FrontEnd()
    run = fe.run_cf32 if fmt == "cf32" else fe.run_u8
    chunks = [data[o:o + step] for o in range(0, len(data), step)]
    for c in chunks[:4]: run(c)
    fe.reset()
    t0 = time.perf_counter(); got = [run(c) for c in chunks]; dt = time.perf_counter() - t0
    got = np.concatenate(got)
    k = min(len(got), len(truth) - 24)
    ok = bool(k > 0 and np.array_equal(got[:k], truth[24:24 + k]))          # the modulator's symbols (parity vs the oracle: tests/)
    print(fmt, "ms/chunk %.4f" % (dt / len(chunks) * 1e3), "chunks", len(chunks), "parity", ok)
    # raw C call timing without the Python wrapper's allocations
    import ctypes as C
    dib = np.empty((1, 400), dtype=np.uint8); nd = (C.c_size_t * 1)()
    fn = fe.L.p25fe_run_cf32 if fmt == "cf32" else fe.L.p25fe_run_u8
    fe.reset()
    arrs = [np.ascontiguousarray(c) for c in chunks]
    t0 = time.perf_counter()
    for a in arrs:
        fn(fe.h, a.ctypes.data_as(C.c_void_p), a.size, dib.ctypes.data_as(C.c_void_p), 400, nd)
    dt = time.perf_counter() - t0
    print(fmt, "raw C ms/chunk %.4f" % (dt / len(chunks) * 1e3))
