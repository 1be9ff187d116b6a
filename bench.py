#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through FM-demod + C4FM slice on MI355X (BASELINE.json metric).

N = 1 (the headline): a "step" is one pass of the hot path (K1 front end -> K2 sync detect -> K3 scan -> K4 slice)
over one resident capture: BASELINE.json configs[1], 1 channel x 600 s of synthetic C4FM IQ at 240 ksps
(1.44e8 cf32 samples, 1.152 GB), generated in HBM before the timed region.  The same JSON line carries an
`extra` array with the other single-GPU configurations (u8 input, config 3 end to end, config 4, config 5 on one GPU),
each timed over its own >= 100 ms region after the headline one (`--no-extra` skips them).

N > 1 (one rank per GPU): the time-sharded step of the C ABI -- p25fe_shard_step of include/p25fe_rccl.h, driven through
ctypes (p25rx_amd/rccl.py), RCCL inside libp25fe_rccl.so: halo send / recv to the right neighbour (hidden behind K1) ->
pass 1 -> all-gather of the 96-byte shard summaries -> device resolve -> pass 2 -> every shard's valid dibits point-to-point
to rank 0, received at their offsets of ONE ordered stream.  torch.distributed (gloo) is the control plane only: the
communicator id, the barriers around the timed region, the max over ranks, the gates.  Default `--scaling weak`: every GPU
holds configs[1]'s 600 s (the N = 1 line and the N > 1 lines are one curve of fixed per-GPU work); `--scaling strong` is
BASELINE.json configs[4] as worded, ONE 3 600 s capture cut N ways.  A plain `python bench.py --gpus N` (no WORLD_SIZE in
the environment) starts the N ranks itself, as children, through torch.distributed.run.  An N > 1 run that cannot produce its number
(rendezvous, RCCL bootstrap, a rank that never arrives: a watchdog, P25FE_BENCH_WATCHDOG_S, default 900 s) still prints ONE line on rank 0:
the contract's keys, value 0, `error` and the `stage` it happened in, and exits non-zero (also when the launcher ends rank 0 with SIGTERM
because another rank failed first).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel (K1) from HIP
events recorded inside the library on the launch stream, and `cpu_baseline` from the CPU oracle timed on this box's
host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3         # fp32 vector peak (MI355X_MICROARCH.md)
# algorithmic bytes per input IQ sample of the fused path = 8 B read + 1 dibit byte per 50
# samples written.  The 0.8 B / sample polyphase baseband that K1 hands to K2 / K4 is this design's INTERMEDIATE, not an
# output: it is traffic, not algorithmic bytes.  (K1 priced alone on what it reads and writes, 8.8 B, is reported beside
# it as `k1_alone`.)
BYTES_PER_SAMPLE = 8.02
BYTES_PER_SAMPLE_U8 = 2.02
BYTES_PER_SAMPLE_K1 = 8.0 + 0.8
BYTES_PER_SAMPLE_K1_U8 = 2.0 + 0.8
# arithmetic of SPEC section 3 per input sample (flops: an fma = 2): decimator 31 taps x 2 components / 5, channel filter
# 41 x 2 / 5, discriminator (4 + division + 8-term Horner + selects ~ 30) / 5, boxcar 10 / 5, u8 -> f32 2 x 2
FLOPS_PER_SAMPLE = (2 * 2 * 31 + 2 * 2 * 41 + 30 + 10) / 5.0
FLOPS_PER_SAMPLE_U8 = FLOPS_PER_SAMPLE + 4.0
PMC_FILES = [os.path.join("profiles", "r06_k1_pmc.json"), os.path.join("profiles", "r05_k1_pmc.json"), os.path.join("profiles", "r04_k1_pmc.json")]
PREWARM_MS = 150.0               # untimed steps before the W warm-up steps: the chip reaches its steady clock / power state


STAGE = ["start"]                # where the run is (N > 1: what a failure line names)


def stage(what):
    STAGE[0] = what


def failure_line(args, why):
    """The one JSON line of a run that could not produce its number: the same keys, value 0, and what went wrong where -- a
    rendezvous or an RCCL bootstrap that fails (or hangs: the watchdog) must leave a record, not silence."""
    return json.dumps({"metric": "IQ Msamples/s through FM-demod+C4FM slice", "value": 0.0, "unit": "Msamples/s", "n_gpus": args.gpus,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                       "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                       "config": {"workload": "FAILED before a number existed"}, "error": why, "stage": STAGE[0]})


def start_watchdog(args, rank):
    """N > 1, rank 0: a run stuck in a collective prints a failure line and ends instead of hanging until somebody's timeout."""
    import threading
    limit = float(os.environ.get("P25FE_BENCH_WATCHDOG_S", "900"))
    done = threading.Event()

    def watch():
        if not done.wait(limit):
            if rank == 0:
                print(failure_line(args, "watchdog: no result after %.0f s" % limit), flush=True)
            sys.stderr.write("bench.py rank %d: watchdog after %.0f s in stage '%s'\n" % (rank, limit, STAGE[0]))
            sys.stderr.flush()
            os._exit(5)
    threading.Thread(target=watch, daemon=True).start()
    return done
GATHER_WORDS = {"root_exact": "point-to-point to rank 0, exactly the valid bytes, received at their offsets", "root": "point-to-point "
                "gather of whole rows to rank 0 + compaction", "all": "all_gather of rows", "none": "nothing"}


def cpu_baseline(iq_host, seconds_label):
    """Time the oracle (scalar C port of the reference structure) on one host core, then on all of them."""
    from oracle import oracle as O
    so = "/tmp/p25fe_oracle_timing_%d.so" % os.getpid()
    kind_flags = "-O3 -march=native"
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "timing", "OUT=" + so])
    except Exception:
        so, kind_flags = None, "-O2 -mfma (prebuilt)"
    O.run_cf32(iq_host[:240000], libpath=so)                      # warm
    t0 = time.perf_counter()
    dib = O.run_cf32(iq_host, libpath=so)
    dt = time.perf_counter() - t0
    out = {"value": round(len(iq_host) / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%s of the same capture (%d IQ samples), oracle/p25fe_oracle.c built %s, 1 thread, %.2f s"
                     % (seconds_label, len(iq_host), kind_flags, dt)}
    # second figure (SURVEY 8d ii): the same port on all host cores, the capture cut into one time shard per thread
    # (pthreads inside the oracle; shard-boundary symbols are not stitched -- a throughput figure only)
    ncores = os.cpu_count() or 1
    if O.run_cf32_mt(iq_host, ncores, libpath=so) >= 0:           # untimed: idle cores take a while to come up
        t0 = time.perf_counter()
        rc = O.run_cf32_mt(iq_host, ncores, libpath=so)
        dt_all = time.perf_counter() - t0
        if rc >= 0:                                               # a failed or partial pass is not a baseline
            out["all_cores"] = {"value": round(len(iq_host) / dt_all / 1e6, 3), "cores": ncores,
                                "note": "same sample as %d independent time shards, one thread each, %.2f s" % (ncores, dt_all)}
    if so and os.path.exists(so):
        os.unlink(so)
    return out, dib


def timed(torch, step, steps, warmup, dist=None, finish=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns seconds (max over ranks).
    `finish` (pipelined steps: FrontEnd.join_dev) runs after the K-th step, inside the timed region; the device-wide
    synchronize that follows covers every stream, so all work of the K steps is complete when the clock stops."""
    for _ in range(warmup):
        step()
    if finish:
        finish()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if finish:
        finish()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def steps_for(ms_guess, min_ms=120.0, lo=5, hi=2000):
    return int(max(lo, min(hi, np.ceil(min_ms / ms_guess))))


def k1_frac(fe, torch, step, n_samples, bytes_per_sample, reps=8):
    """Dominant-kernel time from the library's HIP events (on the launch stream), over a few extra SERIAL steps
    (the extras' ms_per_step is that of pipelined steps; the kernel split is that of the kernels alone)."""
    fe.profile_enable(1)
    for _ in range(reps):
        step()
    fe.join_dev()
    torch.cuda.synchronize()
    kms, ncalls = fe.profile_read()
    fe.profile_enable(False)
    k1 = kms[0] / max(ncalls, 1)
    ach = bytes_per_sample * n_samples / (k1 * 1e-3) / 1e9 if k1 > 0 else 0.0
    return k1, ach, [m / max(ncalls, 1) for m in kms]


def run_extras(torch, dev, args, iq2, truth2):
    """The other single-GPU configurations, each over its own timed region; returns a list of dicts."""
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd, parse_results
    out = []
    pipe = not args.no_pipeline

    def run(fe, x, dib, res):
        return fe.run_dev_pipelined(x, dibits=dib, result=res) if pipe else fe.run_dev(x, dibits=dib, result=res)

    def entry(name, n_samples, ms, k_name, k_ms, bps, ok, **kw):
        ach = bps * n_samples / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        d = {"config": name, "ms_per_step": round(ms, 4), "Msamples_per_s": round(n_samples / ms / 1e3, 1),
             "dominant_kernel": k_name, "kernel_ms": round(k_ms, 4), "achieved_GBps": round(ach, 1),
             "frac": round(ach / HBM_PEAK_GBPS, 4), "parity_gate": bool(ok)}
        d.update(kw)
        out.append(d)

    def gate(dib, res, truth, ch=0):
        nd = int(parse_results(res)[ch]["n_dibits"])
        got = dib[ch, :nd].cpu().numpy()
        k = min(nd, len(truth) - 24)
        return k > 0 and np.array_equal(got[:k], truth[24:24 + k])

    # ---- configs[0]: the reference's own shape -- 1 s of recorded cf32 IQ fed chunk by chunk (16 384 samples = one
    # 32 768-byte read of the dongle, src/consts.rs:6-8) through the host-buffer streaming entry point, state carried
    # across chunks, dibits back on the host after every chunk: PCIe and launch latency included, nothing resident
    n1 = 240000
    host1 = iq2[:n1].cpu().numpy().view(np.complex64).reshape(-1)
    fe = FrontEnd(device=dev.index)
    chunks = [host1[o:o + 16384] for o in range(0, n1, 16384)]
    for c in chunks[:3]:
        fe.run_cf32(c)
    fe.reset()
    t0 = time.perf_counter()
    got1 = [fe.run_cf32(c) for c in chunks]
    dt1 = time.perf_counter() - t0
    got1 = np.concatenate(got1)
    k1_ = min(len(got1), len(truth2) - 24)
    # the same second as RTL-SDR bytes: the reference's own read size, 32 768 bytes per call (src/consts.rs:6)
    u8_1 = c4fm.to_u8(host1)
    fe.reset()
    chunks8 = [u8_1[o:o + 32768] for o in range(0, len(u8_1), 32768)]
    for c in chunks8[:3]:
        fe.run_u8(c)
    fe.reset()
    t0 = time.perf_counter()
    got8 = [fe.run_u8(c) for c in chunks8]
    dt8 = time.perf_counter() - t0
    got8 = np.concatenate(got8)
    k8_ = min(len(got8), len(truth2) - 24)
    out.append({"config": "configs[0]: 1 s cf32 @ 240 ksps in 16 384-sample chunks through the host-buffer streaming API "
                          "(p25fe_run_cf32: pinned zero-copy staging, ONE launch -- K1 + receiver by the last workgroup --, "
                          "one synchronisation per chunk)", "ms_per_chunk": round(dt1 / len(chunks) * 1e3, 4),
                "chunks": len(chunks), "Msamples_per_s": round(n1 / dt1 / 1e6, 2), "realtime_factor": round(1.0 / dt1, 1),
                "parity_gate": bool(k1_ > 0 and np.array_equal(got1[:k1_], truth2[24:24 + k1_])),
                "u8_32768_byte_chunks": {"ms_per_chunk": round(dt8 / len(chunks8) * 1e3, 4), "Msamples_per_s": round(n1 / dt8 / 1e6, 2),
                                         "parity_gate": bool(k8_ > 0 and np.count_nonzero(got8[:k8_] != truth2[24:24 + k8_]) == 0)},
                "note": "latency-bound by construction (one 68 ms chunk per call); the reference needs 1.0x real time; "
                        "timed through the Python wrapper (ctypes + two small NumPy allocations per call)"})
    del fe

    # ---- configs[1] as RTL-SDR u8 pairs: the reference's real input format (src/demod.rs:74-84)
    n = iq2.shape[0]
    fe = FrontEnd(device=dev.index)
    u8 = torch.clamp(torch.round((iq2 + 1.0) * 127.5), 0, 255).to(torch.uint8)
    dib = res = None
    def step_u8():
        nonlocal dib, res
        dib, res = run(fe, u8, dib, res)
    k = steps_for(0.3)
    dt = timed(torch, step_u8, k, 5, finish=fe.join_dev)
    k1, _, _ = k1_frac(fe, torch, lambda: fe.run_dev(u8, dibits=dib, result=res), n, BYTES_PER_SAMPLE_U8)
    tfl = FLOPS_PER_SAMPLE_U8 * n / (k1 * 1e-3) / 1e12 if k1 > 0 else 0.0
    entry("configs[1] as u8 I/Q pairs (the reference's input format), 1 ch x 600 s", n, dt / k * 1e3, "k_frontend<u8>", k1,
          BYTES_PER_SAMPLE_U8, gate(dib, res, truth2), steps=k,
          roofline={"bound": "valu", "achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tfl / VALU_PEAK_TFLOPS, 4), "flops_per_sample": FLOPS_PER_SAMPLE_U8,
                    "note": "2.02 B per sample leaves HBM idle: this format is priced on the fp32 vector pipe (SPEC section 3's "
                            "arithmetic, an fma = 2 flops); instruction mix in profiles/r03_pmc_u8.txt"})
    del u8, fe

    # ---- configs[1] from HOST memory at the speed of the bus (p25fe_run_host_windows): the capture sits in pinned host memory,
    # window k + 1 travels to the GPU while window k is in the kernels and window k - 1's dibits travel back; filter and
    # receiver state cross the window boundaries as they cross a shard's.  Priced against THIS box's host-to-device copy rate
    # (one pinned -> device copy of the same bytes, measured right here); `value` above never includes the bus.
    def host_case(name, x_dev, ref_dib, ref_res, bytes_per_sample):
        host = x_dev.cpu().pin_memory()
        dst = torch.empty_like(x_dev)
        dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0_.record()
        for _ in range(3):
            dst.copy_(host, non_blocking=True)
        e1_.record()
        torch.cuda.synchronize()
        nbytes = host.numel() * host.element_size()
        h2d_gbps = 3 * nbytes / (e0_.elapsed_time(e1_) * 1e-3) / 1e9
        del dst
        fe_ = FrontEnd(device=dev.index)
        got_, st_ = fe_.run_host_windows(host)                     # warm-up: allocates the windows
        best, st_best, wrap_ms = None, None, None
        for _ in range(3):
            fe_.reset()
            t0_ = time.perf_counter()
            got_, st_ = fe_.run_host_windows(host)
            dt_py = time.perf_counter() - t0_
            dt_ = st_["ms_total"] * 1e-3                            # wall time INSIDE the C call (what a C / Rust host sees); the Python
            if best is None or dt_ < best:                          # wrapper adds two NumPy allocations and a 2.9 MB copy on top
                best, st_best, wrap_ms = dt_, st_, dt_py * 1e3
        nd_ = int(parse_results(ref_res)[0]["n_dibits"])
        same = bool(len(got_) == nd_ and np.array_equal(got_, ref_dib[0, :nd_].cpu().numpy()))
        # zero copy for comparison: the resident-capture call on the pinned host memory itself (the kernels read it over the bus)
        zc_ms = None
        try:
            class _Raw:
                pass
            raw = _Raw()
            raw.__cuda_array_interface__ = {"shape": tuple(host.shape), "typestr": "|u1" if host.dtype == torch.uint8 else "<f4",
                                            "version": 2, "data": (int(host.data_ptr()), False)}
            view = torch.as_tensor(raw, device=dev)
            dz, rz = fe_.run_dev(view)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            dz, rz = fe_.run_dev(view, dibits=dz, result=rz)
            torch.cuda.synchronize()
            if int(parse_results(rz)[0]["n_dibits"]) == nd_ and bool(torch.equal(dz[0, :nd_], ref_dib[0, :nd_])):
                zc_ms = (time.perf_counter() - t0_) * 1e3
        except Exception:
            zc_ms = None
        ach_ = nbytes / best / 1e9
        out.append({"config": "configs[1] %s from pinned HOST memory through p25fe_run_host_windows (64 MB windows: H2D copy | K1..K4 | "
                              "dibits back, three streams)" % name, "ms_total": round(best * 1e3, 3),
                    "Msamples_per_s": round(x_dev.shape[0] / best / 1e6, 1), "windows": st_best["n_windows"],
                    "ms_through_the_python_wrapper": round(wrap_ms, 3),
                    "ms_h2d_copies": round(st_best["ms_h2d"], 3), "ms_kernels": round(st_best["ms_compute"], 3),
                    "roofline": {"bound": "pcie", "achieved": round(ach_, 2), "peak": round(h2d_gbps, 2), "unit": "GB/s",
                                 "frac": round(ach_ / h2d_gbps, 4),
                                 "peak_source": "one pinned-host -> device copy of the same %d bytes on this box, 3 repetitions" % nbytes},
                    "zero_copy_single_launch_ms": (round(zc_ms, 3) if zc_ms else None),
                    "parity_gate": same, "gate": "dibits == p25fe_run_dev of the same capture resident in HBM, byte for byte"})
        del host, fe_

    fe = FrontEnd(device=dev.index)
    rd, rr = fe.run_dev(iq2)
    torch.cuda.synchronize()
    host_case("cf32", iq2, rd, rr, 8)
    u8h = torch.clamp(torch.round((iq2 + 1.0) * 127.5), 0, 255).to(torch.uint8)
    rd, rr = fe.run_dev(u8h)
    torch.cuda.synchronize()
    host_case("as u8 I/Q pairs", u8h, rd, rr, 2)
    del fe, rd, rr, u8h

    # ---- configs[1] with tables and constants that are NOT the build's own -- what a pinned build runs: the reference fixes
    # p25_filts' tables at compile time (src/demod.rs:27-29), the library compiles the same kernels for the caller's numbers
    # (hipRTC at p25fe_create, cached).  Other Kaiser designs of the two filters (the signal still decodes: the truth gate
    # applies), 31 / 41 and 64 / 64 taps, cf32 and u8; beside them the generic LDS-tap fallback and the hipRTC product of the
    # build's OWN numbers (same source, same options: the time must be the built-in kernels').
    from scipy import signal as sps
    from p25rx_amd import _lib

    def design(n1, n2):
        h1 = sps.firwin(n1, 11000.0, window=("kaiser", 6.0), fs=240000.0).astype(np.float32)
        h2 = sps.firwin(n2, 6500.0, window=("kaiser", 4.5), fs=48000.0).astype(np.float32)
        return h1.tolist(), h2.tolist()

    def flops_per_sample(t1, t2, is_u8):
        return (2 * 2 * t1 + 2 * 2 * t2 + 30 + 10) / 5.0 + (4.0 if is_u8 else 0.0)

    def oracle_prefix_gate(fe_, x_, kw_, is_u8, n_pre=480000):
        """GPU baseband of the first 2 s == the oracle configured with the same numbers, bit for bit"""
        from oracle import oracle as O
        bb_, nb_ = fe_.demod_dev(x_[:n_pre])
        pre = x_[:n_pre].cpu().numpy()
        d_ = O.Demod(O.make_config(None, **kw_))
        ref_ = d_.feed_u8(pre.reshape(-1)) if is_u8 else d_.feed_cf32(pre.view(np.complex64).reshape(-1))
        return bool(nb_ == len(ref_) and np.array_equal(bb_[0, :nb_].cpu().numpy().view(np.uint32), ref_.view(np.uint32)))

    def oracle_dibit_prefix_gate(fe_, x_, kw_, is_u8, n_pre=480000):
        """... and the dibits of that prefix == the oracle's receiver on the oracle's baseband (for filters whose delay moves
        the frame against the modulator's symbol list)"""
        from oracle import oracle as O
        pre = x_[:n_pre].cpu().numpy()
        cfg_ = O.make_config(None, **kw_)
        d_ = O.Demod(cfg_)
        ref_ = d_.feed_u8(pre.reshape(-1)) if is_u8 else d_.feed_cf32(pre.view(np.complex64).reshape(-1))
        want = O.Recv(cfg_).feed(ref_)[0]
        dib_, res_ = fe_.run_dev(x_[:n_pre])
        nd_ = int(parse_results(res_)[0]["n_dibits"])
        return bool(nd_ == len(want) and np.array_equal(dib_[0, :nd_].cpu().numpy(), want))

    u8_all = torch.clamp(torch.round((iq2 + 1.0) * 127.5), 0, 255).to(torch.uint8)
    variant_name = {0: "built-in immediates", 1: "specialised (hipRTC) immediates", 2: "generic LDS taps"}
    with open(os.path.join(ROOT, "tests", "golden", "spec.json")) as f_:
        rc_taps = json.load(f_)["rc_avg_taps"]
    cases = [("31 / 41 non-default tables", dict(zip(("decim_taps", "chan_taps"), design(31, 41))), _lib.SPECIALIZE_REQUIRE, 31, 41),
             ("64 / 64 non-default tables", dict(zip(("decim_taps", "chan_taps"), design(64, 64))), _lib.SPECIALIZE_REQUIRE, 64, 64),
             ("31 / 41 non-default tables, generic fallback", dict(zip(("decim_taps", "chan_taps"), design(31, 41))), _lib.SPECIALIZE_OFF, 31, 41),
             ("the build's own numbers through hipRTC", {}, _lib.SPECIALIZE_FORCE, 31, 41),
             # ABI 5: MovingAverage::new(10) replaced by the raised-cosine filter north_star names (41 taps, tools/gen_spec.py) and
             # the decimator on another phase: the last two constructor numbers of DemodTask::new as data
             ("raised-cosine post-discriminator filter (41 taps, avg_taps) + decim_phase 2", dict(avg_taps=rc_taps, decim_phase=2),
              _lib.SPECIALIZE_REQUIRE, 31, 41)]
    for label, kw, spz, t1, t2 in cases:
        for is_u8 in (False, True):
            x = u8_all if is_u8 else iq2
            t_c = time.perf_counter()
            fe = FrontEnd(device=dev.index, specialize=spz, **kw)
            t_create = time.perf_counter() - t_c
            dib = res = None
            def step_ct():
                nonlocal dib, res
                dib, res = run(fe, x, dib, res)
            k = steps_for(0.3)
            dt = timed(torch, step_ct, k, 5, finish=fe.join_dev)
            k1, _, _ = k1_frac(fe, torch, lambda: fe.run_dev(x, dibits=dib, result=res), n, BYTES_PER_SAMPLE_U8 if is_u8 else BYTES_PER_SAMPLE)
            rc_case = "avg_taps" in kw
            if rc_case:
                ok = oracle_prefix_gate(fe, x, kw, is_u8) and oracle_dibit_prefix_gate(fe, x, kw, is_u8)
            else:
                ok = gate(dib, res, truth2) and oracle_prefix_gate(fe, x, kw, is_u8)
            fl = flops_per_sample(t1, t2, is_u8) + (2.0 * (len(kw["avg_taps"]) - 10) / 5.0 if rc_case else 0.0)
            tfl = fl * n / (k1 * 1e-3) / 1e12 if k1 > 0 else 0.0
            bps = BYTES_PER_SAMPLE_U8 if is_u8 else BYTES_PER_SAMPLE
            ach = bps * n / (k1 * 1e-3) / 1e9 if k1 > 0 else 0.0
            roof = ({"bound": "valu", "achieved": round(tfl, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tfl / VALU_PEAK_TFLOPS, 4), "flops_per_sample": fl} if is_u8 else
                    {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                     "valu_TFLOPs": round(tfl, 2), "flops_per_sample": fl})
            entry("configs[1] as %s, %s" % ("u8 I/Q pairs" if is_u8 else "cf32", label), n, dt / k * 1e3,
                  "k_frontend<%s> %s" % ("u8" if is_u8 else "cf32", variant_name[fe.kernel_variant]), k1, bps, ok, steps=k,
                  kernel_variant=fe.kernel_variant, create_s=round(t_create, 2), roofline=roof,
                  gate=("baseband AND dibits of the first 2 s == oracle with the same numbers, bit for bit" if rc_case else
                        "dibits == modulator symbols AND baseband of the first 2 s == oracle with the same numbers, bit for bit"))
            del fe, dib, res
    del u8_all

    # ---- configs[1] with the tracking symbol clock (SPEC 3.8b / 3.8c; north_star's "symbol-clock interpolator"): the general receiver
    # The capture is generated with a sample clock 150 ppm fast (the modulator's phase trajectory read at m / (1 + 150e-6): sync-to-sync
    # intervals are 8 641.3 baseband samples), so that the receiver runs its general D / N arithmetic with non-zero interpolation
    # phases (on the exact capture every interval is 10 N and the clock degenerates to the fixed stride's fast path) and the fixed
    # stride walks 1.3 samples off the eye per frame.  The modulator's symbols are the truth for all three clocks: the line reports
    # their symbol errors side by side.  (Rounds 2 - 4 dropped one IQ sample in 6 667 instead: 0.2-sample jumps no clock model follows,
    # which cost 45 symbols of 2.88 M -- inside the NEXT sync word's last symbols of scattered frames, not in the first frame.)
    iq_ppm, truth_ppm = c4fm.synth_torch(n, seed=1003, device=dev, snr_db=30.0, clock_ppm=150.0)
    n_ppm = n

    def sym_errors(dib_, res_, where=False):
        nd_ = int(parse_results(res_)[0]["n_dibits"])
        got_ = dib_[0, :nd_].cpu().numpy()
        k_ = min(nd_, len(truth_ppm) - 24)
        bad_ = np.nonzero(got_[:k_] != truth_ppm[24:24 + k_])[0]
        return (bad_, k_) if where else (int(len(bad_)), k_)

    fe0 = FrontEnd(device=dev.index)
    d0, r0 = fe0.run_dev(iq_ppm)
    torch.cuda.synchronize()
    err_fixed, k_fixed = sym_errors(d0, r0)
    del fe0, d0, r0
    # symbol_clock = 1, the causal rule -- what EVERY entry point can run (streaming chunks, host windows, time shards): timed on the same
    # capture, its own line, so that the mode-2 line below has its comparable beside it
    fe1 = FrontEnd(device=dev.index, symbol_clock=1)
    d1 = r1 = None
    def step_causal():
        nonlocal d1, r1
        d1, r1 = run(fe1, iq_ppm, d1, r1)
    k = steps_for(0.30)
    dt1 = timed(torch, step_causal, k, 5, finish=fe1.join_dev)
    k1c, _, kmsc = k1_frac(fe1, torch, lambda: fe1.run_dev(iq_ppm, dibits=d1, result=r1), n, BYTES_PER_SAMPLE)
    bad_causal, k_causal = sym_errors(d1, r1, where=True)
    err_causal = int(len(bad_causal))
    entry("configs[1] with a 150 ppm sample clock and symbol_clock = 1 (tracking, CAUSAL: period from the last two sync words, 4-tap "
          "interpolated instants; SPEC 3.8b) -- the rule every entry point runs, streaming / windows / time shards included",
          n, dt1 / k * 1e3, "k_frontend<cf32>", k1c, BYTES_PER_SAMPLE,
          k_causal > 2800000 and err_causal <= k_causal // 10000 and err_causal < err_fixed, steps=k,
          receiver_ms={"k_detect<general>": round(kmsc[1], 4), "k_scan_tiles_g (group scans + the range's walk)": round(kmsc[2], 4),
                       "k_scan_g_groups + k_slice_g": round(kmsc[3], 4)},
          symbol_errors={"tracking_causal (symbol_clock 1, timed)": err_causal, "fixed_stride_same_capture": err_fixed, "of": k_causal},
          gate="symbol errors vs the modulator over the whole capture: <= 0.01 % and fewer than the fixed stride's")
    del fe1, d1, r1
    # timed: symbol_clock = 2 -- the tracking clock, and the resident call slices the first frame of a lock run with the period the NEXT
    # sync word confirms and starts every detection's instants from its refined position s + f / 4 (SPEC 3.8c: the slicer by detection, k_ev_collect / k_ev_count / k_ev_scan / k_ev_slice behind the general receiver's scan)
    fe = FrontEnd(device=dev.index, symbol_clock=2)
    dib = res = None
    def step_trk():
        nonlocal dib, res
        dib, res = run(fe, iq_ppm, dib, res)
    k = steps_for(0.35)
    dt = timed(torch, step_trk, k, 5, finish=fe.join_dev)
    k1, _, kms = k1_frac(fe, torch, lambda: fe.run_dev(iq_ppm, dibits=dib, result=res), n_ppm, BYTES_PER_SAMPLE)
    a_out = parse_results(res)[0]["anchor_out"]
    bad_trk, k_trk = sym_errors(dib, res, where=True)
    err_trk = int(len(bad_trk))
    first_causal, first_trk = int(np.count_nonzero(bad_causal < 864)), int(np.count_nonzero(bad_trk < 864))
    entry("configs[1] with a 150 ppm sample clock and symbol_clock = tracking + first-frame re-slice (period from sync word to sync word, "
          "4-tap interpolated instants; SPEC 3.8b / 3.8c)", n_ppm, dt / k * 1e3, "k_frontend<cf32>", k1, BYTES_PER_SAMPLE,
          k_trk > 2800000 and err_trk == 0 and first_causal >= 1 and err_causal <= k_trk // 10000 and err_causal < err_fixed, steps=k,
          receiver_ms={"k_detect<general>": round(kms[1], 4), "k_scan_tiles_g (group scans + the range's walk)": round(kms[2], 4),
                       "k_scan_g_groups + k_ev_collect + k_ev_count + k_ev_scan + k_ev_slice": round(kms[3], 4)},
          entry_points_with_3_8c="p25fe_run_dev, p25fe_run_dev_pipelined (timed here), p25fe_slice_dev -- the calls that hold the whole range; p25fe_slice, "
                                 "p25fe_run_u8 / _cf32, p25fe_run_host_windows and the time-shard passes cannot run it and return P25FE_ERR_ARG on such a handle "
                                 "unless it was created with P25FE_CLOCK_CAUSAL_OK (then: the symbol_clock = 1 line above)",
          last_period="%d / %d" % (int(a_out["period_d"]), int(a_out["period_n"])),
          symbol_errors={"tracking_reslice (symbol_clock 2, timed)": err_trk, "tracking_causal (symbol_clock 1)": err_causal,
                         "in_the_first_frame": {"reslice": first_trk, "causal": first_causal},
                         "fixed_stride_same_capture": err_fixed, "of": k_trk},
          gate="symbol errors vs the modulator over the whole capture: 0 under symbol_clock = 2 (first frame re-sliced, instants from the refined "
               "sync position); the causal rule of symbol_clock = 1 loses some in the first frame and <= 0.01 % in all, the fixed stride far more")
    del fe, iq_ppm

    # ---- configs[2]: 2.4 Msps front end, 60 s = 1.44e8 samples -> stage 0 (10:1, 80 taps) -> K1..K4
    n240 = 60 * 240000
    iq240, truth3 = c4fm.synth_torch(n240, seed=31, device=dev, snr_db=30.0)
    wide = iq240.repeat_interleave(10, dim=0).contiguous()          # zero-order hold: images fall in K0's stop band
    del iq240
    fe = FrontEnd(device=dev.index)
    nar = dib = res = None
    def step_c3():
        nonlocal nar, dib, res
        nar, no = fe.predecim_dev(wide, out=nar)
        dib, res = run(fe, nar[:, :no], dib, res)
    k = steps_for(0.36)
    dt = timed(torch, step_c3, k, 5, finish=fe.join_dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        nar, _ = fe.predecim_dev(wide, out=nar)                     # K0 alone, on torch's current stream = the launch stream
    e1.record()
    torch.cuda.synchronize()
    k0_ms = e0.elapsed_time(e1) / 8
    nd = int(parse_results(res)[0]["n_dibits"])
    al = c4fm.align_dibits(dib[0, :nd].cpu().numpy(), truth3)
    entry("configs[2]: 2.4 Msps x 60 s -> 10:1 pre-decimator -> FIR + FM + slice, end to end", wide.shape[0], dt / k * 1e3,
          "k_predecim", k0_ms, 8.002, al is not None and al[3] == 0 and al[2] > 287000, steps=k)
    # ---- polyphase channeliser: the same 60 s wideband capture -> 192 channel streams at 240 ksps
    chz = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    chz, no = fe.channelise_dev(wide, out=chz)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        chz, no = fe.channelise_dev(wide, out=chz)
    e1.record()
    torch.cuda.synchronize()
    k6_ms = e0.elapsed_time(e1) / 4
    k6_bytes = 8.0 + 192 * 8 / 10.0                                 # 8 B read + 192 channels x 8 B per 10 input samples
    ach6 = k6_bytes * wide.shape[0] / (k6_ms * 1e-3) / 1e9
    out.append({"config": "polyphase channeliser: 2.4 Msps x 60 s -> 192 x 240 ksps (k_channelise alone)", "kernel_ms": round(k6_ms, 4),
                "achieved_GBps": round(ach6, 1), "frac": round(ach6 / HBM_PEAK_GBPS, 4), "bytes_per_input_sample": k6_bytes,
                "parity_gate": bool(torch.allclose(chz[0, :no], nar[0, :no], atol=2e-5)),
                "note": "store-bound (154 of 161.6 B per input sample are output); gate: channel 0 equals the pre-decimator's stream"})
    del wide, nar, fe, chz

    # ---- configs[3]: 256 independent channels x 60 s, channel-major (29.5 GB)
    C, n4 = 256, 60 * 240000
    iq4 = torch.empty((C, n4, 2), dtype=torch.float32, device=dev)
    truth4 = None
    for c in range(C):
        _, t = c4fm.synth_torch(n4, seed=2000 + c, device=dev, snr_db=30.0, out=iq4[c])
        if c == C - 1:
            truth4 = t
    fe = FrontEnd(n_channels=C, device=dev.index)
    dib = res = None
    def step_c4():
        nonlocal dib, res
        dib, res = run(fe, iq4, dib, res)
    k = steps_for(7.5)
    dt = timed(torch, step_c4, k, 2, finish=fe.join_dev)
    k1, _, kms = k1_frac(fe, torch, lambda: fe.run_dev(iq4, dibits=dib, result=res), C * n4, BYTES_PER_SAMPLE, reps=3)
    entry("configs[3]: 256 channels x 60 s, channel-major batch", C * n4, dt / k * 1e3, "k_frontend<cf32>", k1,
          BYTES_PER_SAMPLE, gate(dib, res, truth4, C - 1), steps=k,
          receiver_share=round((kms[1] + kms[2] + kms[3]) / max(sum(kms), 1e-9), 4))
    del iq4, fe, dib, res

    # ---- configs[4] on ONE GPU: 3 600 s x 1 channel (6.9 GB)
    n5 = 3600 * 240000
    iq5, truth5 = c4fm.synth_torch(n5, seed=77, device=dev, snr_db=30.0)
    fe = FrontEnd(device=dev.index)
    dib = res = None
    def step_c5():
        nonlocal dib, res
        dib, res = run(fe, iq5, dib, res)
    k = steps_for(1.9)
    dt = timed(torch, step_c5, k, 2, finish=fe.join_dev)
    k1, _, _ = k1_frac(fe, torch, lambda: fe.run_dev(iq5, dibits=dib, result=res), n5, BYTES_PER_SAMPLE, reps=3)
    entry("configs[4] on one GPU: 3 600 s x 1 channel, single pass", n5, dt / k * 1e3, "k_frontend<cf32>", k1,
          BYTES_PER_SAMPLE, gate(dib, res, truth5), steps=k)
    return out


def bench_channels(args, torch, dist, world, rank, local, dev, staged):
    """configs[3] over N GPUs: rank r owns a contiguous block of the channel batch (strong: the batch is fixed).  A step is
    FrontEnd.run_dev over the local block -- the data path has no collective; the per-channel summaries are all-gathered
    once after the timed region (the consumer's table), and checked."""
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd.sharding import ChannelShard, HostStagedComm
    secs = 60.0 if args.seconds is None else args.seconds
    n = int(round(secs * 240000)) // 8 * 8
    cs = ChannelShard(rank, world, args.channels, dist,
                      comm=HostStagedComm(dist, rank, world) if (staged and world > 1) else None)
    iq = torch.empty((cs.n_local, n, 2), dtype=torch.float32, device=dev)
    truths = []
    for c in range(cs.n_local):
        _, t = c4fm.synth_torch(n, seed=2000 + cs.c0 + c, device=dev, snr_db=30.0, out=iq[c])
        truths.append(t)
    fe = FrontEnd(n_channels=cs.n_local, device=local)
    dib = res = None

    def step():
        nonlocal dib, res
        dib, res = cs.step(fe, iq, dibits=dib, result=res)

    dt = timed(torch, step, args.steps, args.warmup, dist)
    k1, ach, _ = k1_frac(fe, torch, step, cs.n_local * n, BYTES_PER_SAMPLE, reps=3)
    cdev = torch.device("cpu") if staged else dev
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    r = parse_results(res)
    ok = True
    for c in (0, cs.n_local - 1):
        nd = int(r[c]["n_dibits"])
        got = dib[c, :nd].cpu().numpy()
        k = min(nd, len(truths[c]) - 24)
        ok = ok and bool(k > 0 and np.array_equal(got[:k], truths[c][24:24 + k]))
    table = cs.gather_results(torch, res)                              # [channels, sizeof(p25fe_result_t)] on every rank
    allr = parse_results(table)
    ok_table = bool(len(allr) == args.channels and all(int(x["n_dibits"]) > 0 for x in allr)
                    and np.array_equal(allr[cs.c0:cs.c1]["n_dibits"], r["n_dibits"]))
    if dist:
        okt = torch.tensor([1 if ok else 0, 1 if ok_table else 0], device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok, ok_table = bool(okt[0].item()), bool(okt[1].item())
    if rank == 0:
        total = float(n) * args.channels * args.steps
        print(json.dumps({
            "metric": "IQ Msamples/s through FM-demod+C4FM slice", "value": round(total / dt / 1e6, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: %d independent channels x %.0f s @ 240 ksps, channel-major, %d per rank (blocks)"
                                   % (args.channels, secs, cs.n_local),
                       "sharding": "channel blocks, no collective on the data path; summaries all-gathered after the timed region"
                                   + (" (TEST HOOK: gloo through host copies, all ranks on one GPU)" if staged and world > 1 else ""),
                       "parity_gate": "dibits == modulator symbols (first and last local channel of every rank): %s" % ok,
                       "table_gate": "gathered per-channel table complete and in channel order: %s" % ok_table},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "k_frontend<cf32>",
                         "kernel_ms": round(k1, 4), "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * cs.n_local * n}}))
    if dist:
        dist.destroy_process_group()
    if not (ok and ok_table) and world == 1:
        sys.exit(3)


def shard_evidence(ss, dist, world):
    """What the LIBRARY says about the job, from every rank (p25fe_shard_info gathered over gloo): did RCCL see N ranks, on N different
    GPUs, which communicator layout does the pipelined step run.  None of it comes from the launcher's environment."""
    mine = ss.info()
    infos = [None] * world
    dist.all_gather_object(infos, mine)
    buses = [i["pci_bus_id"] for i in infos]
    return {"rccl_ranks": [i["rccl_ranks"] for i in infos], "rccl_rank_of_rank": [i["rccl_rank"] for i in infos],
            "device_of_rank": [i["device"] for i in infos], "pci_bus_id_of_rank": buses, "distinct_gpus": len(set(buses)),
            "communicators": sorted(set(i["comms"] for i in infos)), "pipe_layout": sorted(set(i["pipe_layout"] for i in infos)),
            "layout_words": {3: "three communicators: halo / summaries / dibit rows each their own (two stages at K1 boundaries)",
                             1: "one communicator: everything behind K1 in step order on the receive stream",
                             0: "no communicator (TEST HOOK: shared memory)"}.get(mine["comms"], "?"),
            "staged_test_hook": bool(mine["staged"]), "broken": any(i["broken"] for i in infos),
            "source": "p25fe_shard_info on every rank: ncclCommCount / ncclCommUserRank of the step's communicator, hipDeviceGetPCIBusId "
                      "of the handle's device, the layout agreed by all-reduce in p25fe_shard_create"}


def strong_row(torch, dist, args, world, rank, local, dev, staged, steps, total_s):
    """BASELINE.json configs[4] AS WORDED, as one more timed region of an N > 1 run: ONE 3 600 s capture cut into N contiguous time
    shards (--scaling strong), the same pipelined step, its own gates.  Every rank runs it (the steps hold collectives); rank 0 gets the row."""
    from p25rx_amd import c4fm, rccl
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd._lib import RESULT_DTYPE
    n = int(round(total_s * 240000)) // world
    n -= n % 8
    fe = FrontEnd(device=local)
    halo = fe.shard_halo()
    buf = torch.zeros((halo + n, 2), dtype=torch.float32, device=dev)
    _, truth = c4fm.synth_torch(n, seed=2000 + rank, device=dev, snr_db=30.0, out=buf[halo:])
    result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    boot = [None]
    if rank == 0:
        boot[0] = ("/p25fe_bench_s_%d" % os.getpid()) if staged else rccl.unique_id()
    dist.broadcast_object_list(boot, src=0)
    if staged:
        os.environ["P25FE_SHARD_SHM"] = boot[0]
    ss = rccl.ShardStep(fe, rank, world, n, None if staged else boot[0])
    ss.prepare(dev)
    dibits = torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device=dev)
    pipe = not args.no_pipeline and not staged

    def step():
        ss.step(buf, dibits, result, gather=args.gather, pipelined=pipe)
    dt = timed(torch, step, steps, 8 if not staged else 1, dist, finish=fe.join_dev)
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    r = parse_results(result)[0]
    nd = int(r["n_dibits"])
    got = dibits[0, :nd].cpu().numpy()
    ok = False
    if int(r["first_event"]) >= 0:
        first = nd - int(r["n_dibits_after_first"])
        for j in (24, 24 + 864, 24 + 2 * 864):
            k = min(nd - first, len(truth) - j)
            if k > 0 and np.array_equal(got[first:first + k], truth[j:j + k]):
                ok = True
                break
    gather_ok = None
    if args.gather != "none":
        off = ss.offsets()
        d_stream = ss.stream(torch, dev, int(off[world])) if (rank == 0 or args.gather == "all") else None
        w = torch.arange(1, nd + 1, dtype=torch.int64, device=dev) % 65521
        mine = dibits[0, :nd].to(torch.int64)
        sig = torch.stack([torch.tensor(nd, dtype=torch.int64, device=dev), mine.sum(), (mine * w).sum()]).cpu()
        sigs = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sigs, sig)
        gather_ok = True
        if rank == 0 or args.gather == "all":
            for r_ in range(world):
                seg = d_stream[int(off[r_]):int(off[r_ + 1])].to(torch.int64)
                wr = torch.arange(1, seg.numel() + 1, dtype=torch.int64, device=dev) % 65521
                gather_ok = gather_ok and [seg.numel(), int(seg.sum().item()), int((seg * wr).sum().item())] == [int(x) for x in sigs[r_].tolist()]
    okt = torch.tensor([1 if ok else 0, 1 if gather_ok in (None, True) else 0])
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    ok, gather_ok = bool(okt[0].item()), (None if args.gather == "none" else bool(okt[1].item()))
    ev = shard_evidence(ss, dist, world)
    comm = None if staged else {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in ss.comm_ms().items()}
    row = None
    if rank == 0:
        row = {"config": "configs[4] as worded: ONE %.0f s capture (%d samples, %.3f GB) cut into %d contiguous time shards of %.1f s "
                         "(--scaling strong), pipelined shard step, gather %s" % (total_s, n * world, n * world * 8 / 1e9, world, n / 240000.0, ss.gather_ran()),
               "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
               "value": round(float(n) * world * steps / dt / 1e6, 1), "unit": "Msamples/s",
               "parity_gate": "every shard's dibits from its first own sync word on == the modulator's symbols: %s" % ok,
               "gather_gate": (None if gather_ok is None else "gathered stream holds every shard at its resolved offset: %s" % gather_ok),
               "rccl": ev, "comm_ms_per_step": comm,
               "whole_step": {"algorithmic_bytes_per_sample": 8.02, "achieved_GBps": round(8.02 * n * world * steps / dt / 1e9, 1),
                              "frac_of_peak_x_gpus": round(8.02 * n * steps / dt / 1e9 / HBM_PEAK_GBPS, 4)}}
    ss.close()
    del buf, dibits
    torch.cuda.empty_cache()
    return row, (ok and gather_ok is not False)


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE): start the N ranks as CHILD processes -- one per GPU,
    through torch.distributed.run exactly as the driver does -- before this process has touched HIP (it never does), relay
    rank 0's JSON line and exit with the children's code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in pr.stdout.decode(errors="replace").splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    sys.exit(pr.returncode if pr.returncode else (0 if js else 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (default: a ~130 ms timed region at 0.32 ms/step)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=None,
                    help="capture length: per GPU at N = 1 / --scaling weak (default 600 = configs[1]), TOTAL at N > 1 strong "
                         "(default 3600 = configs[4])")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak",
                    help="N > 1 only.  weak (default): every GPU holds configs[1]'s 600 s, the capture is N x 600 s cut into N time "
                         "shards -- per-GPU work is that of the N = 1 line, so the per-N values form one curve; strong: configs[4] "
                         "as BASELINE.json words it, ONE 3 600 s capture cut N ways")
    ap.add_argument("--workload", choices=["time", "channels"], default="time",
                    help="N > 1: 'time' = configs[4], one capture cut into time shards (default); 'channels' = configs[3], "
                         "a 256-channel batch cut into channel blocks (no communication on the data path)")
    ap.add_argument("--channels", type=int, default=256, help="--workload channels: size of the batch")
    ap.add_argument("--gather", choices=["root_exact", "root", "all", "none"], default="root",
                    help="N > 1: the dibit shards go to rank 0 as whole rows + a compaction pass (root, the default: no host wait "
                         "anywhere in the step), as exactly their valid bytes received at their offsets of the ordered stream "
                         "(root_exact: one host wait for the offsets per step), all-gathered, or stay sharded (diagnostics)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N = 1: strictly serial steps (K1 -> K2 -> K3 -> K4 on one stream) instead of p25fe_run_dev_pipelined")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configurations (N = 1: the other single-GPU configs; N > 1 weak: the strong row)")
    ap.add_argument("--strong-seconds", type=float, default=3600.0,
                    help="N > 1, --scaling weak: total length of the capture of the extra configs[4]-as-worded (strong) row")
    ap.add_argument("--cpu-seconds", type=float, default=600.0, help="length of the capture prefix timed on the CPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)                                 # (before torch is imported: this process never touches the GPU)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return run(args)
    rank = int(os.environ.get("RANK", "0"))
    finished = start_watchdog(args, rank)
    if rank == 0:
        # torch.distributed.run ends the surviving ranks with SIGTERM when another rank fails first: rank 0 still leaves its line.  A
        # Python-level handler alone would not run while the main thread sits inside a collective or a rendezvous: the C-level handler
        # (whichever thread the signal lands on) writes to a wake-up pipe, and a thread blocked on that pipe prints the line and ends.
        import signal
        import threading
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.signal(signal.SIGTERM, lambda signum, frame: None)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

        def on_term():
            os.read(rfd, 1)
            if not finished.is_set():
                print(failure_line(args, "terminated by the launcher (SIGTERM): another rank failed first -- see its traceback on stderr"), flush=True)
            os._exit(6)
        threading.Thread(target=on_term, daemon=True).start()
    try:
        run(args)
    except SystemExit:
        raise
    except BaseException as e:                                     # N > 1: rank 0 leaves a line that says where
        import traceback
        traceback.print_exc()
        if rank == 0:
            print(failure_line(args, "%s: %s" % (type(e).__name__, e)), flush=True)
        os._exit(4)                                                # (not sys.exit: a rank stuck in a collective's teardown would hang the others)
    finally:
        finished.set()


def run(args):
    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd, parse_results
    from p25rx_amd._lib import RESULT_DTYPE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    staged = bool(os.environ.get("P25FE_BENCH_HOST_STAGED"))     # TEST HOOK: all ranks on GPU 0, collectives through gloo + CPU
    if staged:                                                   # copies (RCCL refuses two ranks on one GPU); never the product path
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: gloo must not try to resolve the container's hostname
        stage("process group (gloo / nccl rendezvous)")
        if staged or args.workload == "time":
            # time shards: the DATA path is libp25fe_rccl.so (RCCL inside the C ABI, p25rx_amd/rccl.py); torch.distributed is
            # the control plane only (communicator id, barrier, max over ranks, gates) and runs on gloo
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.workload == "channels":
        return bench_channels(args, torch, dist, world, rank, local, dev, staged)
    strong = world > 1 and args.scaling == "strong"
    if strong:
        total_s = 3600.0 if args.seconds is None else args.seconds
        n = int(round(total_s * 240000)) // world
    else:
        total_s = 600.0 if args.seconds is None else args.seconds
        n = int(round(total_s * 240000))
    n -= n % 8                                                     # shard cut points stay 16-B aligned
    fe = FrontEnd(device=local)
    halo = fe.shard_halo()
    # resident capture: [halo | owned shard]; the halo region is filled by the neighbour exchange
    buf = torch.zeros((halo + n, 2), dtype=torch.float32, device=dev)
    iq = buf[halo:]
    _, truth = c4fm.synth_torch(n, seed=1000 + rank, device=dev, snr_db=30.0, out=iq)
    torch.cuda.synchronize()

    cap = FrontEnd.dibit_cap(n)
    result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)

    if world == 1:
        dibits = torch.empty((1, cap), dtype=torch.uint8, device=dev)

        if args.no_pipeline:
            def step():
                fe.run_dev(iq, dibits=dibits, result=result)
        else:
            # steps are pipelined two deep inside the library: the receive kernels (K2-K4, three short latency-bound
            # launches) of step i run on the handle's own stream and overlap K1 of step i + 1 -- DemodTask and RecvTask
            # are two threads joined by a channel in the reference too.  All of it completes inside the timed region.
            def step():
                fe.run_dev_pipelined(iq, dibits=dibits, result=result)
    else:
        # the product's N > 1 step: p25fe_shard_create / p25fe_shard_step of include/p25fe_rccl.h through ctypes -- the calls a
        # Rust host binds.  Bootstrap: rank 0's 128-byte id (or, TEST HOOK, the name of a shared-memory object) over gloo.
        from p25rx_amd import rccl
        boot = [None]
        if rank == 0:
            boot[0] = ("/p25fe_bench_%d" % os.getpid()) if staged else rccl.unique_id()
        stage("unique id broadcast")
        dist.broadcast_object_list(boot, src=0)
        if staged:
            os.environ["P25FE_SHARD_SHM"] = boot[0]
        stage("p25fe_shard_create (ncclCommInitRank, ncclCommSplit x 2, agreement all-reduce)")
        ss = rccl.ShardStep(fe, rank, world, n, None if staged else boot[0])
        ss.prepare(dev)                                            # once per stream: the side stream off this stream's hardware queue
        dibits = torch.zeros((1, ss.dibit_cap), dtype=torch.uint8, device=dev)

        # steps one behind the other, like the N = 1 line: only K1's main launch on this stream, everything behind it (detection,
        # scan, summary all-gather, slicer, dibit gather, compaction) on the handle's receive stream beside the NEXT step's K1
        # (p25fe_shard_step_pipelined; all of it completes inside the timed region: join + synchronise).  --no-pipeline: the plain step.
        shard_pipe = not args.no_pipeline and not staged

        def step():
            ss.step(buf, dibits, result, gather=args.gather, pipelined=shard_pipe)

    # untimed pre-warm (not part of W or K): the chip needs ~100 ms of this load to settle at its power-limited clock; the
    # driver's `--steps 20 --warmup 5` is a 6 ms timed region behind 1.5 ms of warm-up and otherwise times the clock ramp
    stage("pre-warm steps")
    t_pw = time.perf_counter()
    n_prewarm = 0
    if world == 1:
        while (time.perf_counter() - t_pw) * 1e3 < PREWARM_MS:
            for _ in range(16):
                step()
            fe.join_dev()
            torch.cuda.synchronize()
            n_prewarm += 16
    elif not staged:
        n_prewarm = 256                                           # N > 1: the same count on every rank (the steps hold collectives)
        for _ in range(n_prewarm):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fe.join_dev()
    torch.cuda.synchronize()
    # K1's begin / end events ride on its own dispatch (hipExtLaunchKernelGGL: no extra packet between two kernels) --
    # on every step when there are at most 64 of them (the ring holds 64 slots), otherwise on every 8th, spread over the run
    fe.profile_enable(2 if (world > 1 or args.steps <= 64) else 3)
    stage("timed steps")
    dt = timed(torch, step, args.steps, 0, dist, finish=fe.join_dev)
    stage("after the timed steps (gates, extras)")
    kms, ncalls = fe.profile_read()
    comm_ms = None
    if world > 1 and not staged:
        # where a step's time goes between the kernels: the library's own events around the three exchanges (last <= 64 steps)
        comm_ms = {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in ss.comm_ms().items()}
    # per-kernel split of the other kernels: a few extra steps OUTSIDE the timed region with events around every kernel
    fe.profile_enable(1)
    n_extra_steps = min(args.steps, 8)
    for _ in range(n_extra_steps):
        if world == 1:
            fe.run_dev(iq, dibits=dibits, result=result)            # serial form: the split is that of the kernels alone
        else:
            step()
    fe.join_dev()
    torch.cuda.synchronize()
    kms_all, _ = fe.profile_read()
    fe.profile_enable(False)
    serial_ms = None
    if world == 1 and not args.no_pipeline:
        # the strictly serial form of the same step (K1 -> K2 -> K3 -> K4 on one stream), for comparison; not `value`
        ks = min(args.steps, 100)
        t0 = time.perf_counter()
        for _ in range(ks):
            fe.run_dev(iq, dibits=dibits, result=result)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t0) / ks * 1e3
    copy_gbps = sum_gbps = None
    if world == 1:
        # context for the roofline: what a plain device-to-device copy of the same capture moves on THIS box, right now
        # (read + write bytes / time over 30 back-to-back copies, torch's copy kernel) -- the practical ceiling of a
        # streaming kernel, against which K1's measured traffic is set below; not part of any timed region above
        try:
            dst = torch.empty_like(iq)
            for _ in range(3):
                dst.copy_(iq)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                dst.copy_(iq)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 2.0 * iq.numel() * 4 * 30 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del dst
            for _ in range(3):
                iq.sum()
            e0.record()
            for _ in range(30):
                iq.sum()
            e1.record()
            torch.cuda.synchronize()
            sum_gbps = iq.numel() * 4 * 30 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception:
            copy_gbps = sum_gbps = None
    cdev = torch.device("cpu")                                   # (the control plane of the time workload is gloo)
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # correctness gate on what the timed steps produced: demod(mod(d)) == d for this rank's shard
    r = parse_results(result)[0]
    nd = int(r["n_dibits"])
    got = dibits[0, :nd].cpu().numpy()
    ok = True
    gather_ok = None
    if world == 1:
        k = min(nd, len(truth) - 24)
        ok = bool(k > 0 and np.array_equal(got[:k], truth[24:24 + k]))
    else:
        # a shard starts mid-stream: its dibits from its first own sync word on must equal the modulator's symbols
        # after a sync word (normally the shard's first; the next two frames are accepted in case the filter
        # transient at the shard boundary costs the first detection)
        ok = False
        if int(r["first_event"]) >= 0:
            first = nd - int(r["n_dibits_after_first"])
            for j in (24, 24 + 864, 24 + 2 * 864):
                k = min(nd - first, len(truth) - j)
                if k > 0 and np.array_equal(got[first:first + k], truth[j:j + k]):
                    ok = True
                    break
        if args.gather != "none":
            # checksum of checksums: every rank's (length, byte sum, position-weighted sum) of its own dibits must equal
            # what rank 0 finds at that shard's resolved offset in the gathered, ordered stream
            if os.environ.get("P25FE_BENCH_INJECT_FAULT") == "gather" and rank == world - 1 and nd > 0:
                dibits[0, 0] ^= 1                                     # TEST HOOK: the gate must see it and the run must exit non-zero
            off = ss.offsets()
            d_stream = ss.stream(torch, dev, int(off[world])) if (rank == 0 or args.gather == "all") else None
            w = torch.arange(1, nd + 1, dtype=torch.int64, device=dev) % 65521
            mine = dibits[0, :nd].to(torch.int64)
            sig = torch.stack([torch.tensor(nd, dtype=torch.int64, device=dev), mine.sum(), (mine * w).sum()]).to(cdev)
            sigs = [torch.zeros(3, dtype=torch.int64, device=cdev) for _ in range(world)]
            dist.all_gather(sigs, sig)
            gather_ok = True
            if rank == 0 or args.gather == "all":
                for r_ in range(world):
                    seg = d_stream[int(off[r_]):int(off[r_ + 1])].to(torch.int64)
                    wr = torch.arange(1, seg.numel() + 1, dtype=torch.int64, device=dev) % 65521
                    got_sig = [seg.numel(), int(seg.sum().item()), int((seg * wr).sum().item())]
                    gather_ok = gather_ok and got_sig == [int(x) for x in sigs[r_].tolist()]
    if dist:
        okt = torch.tensor([1 if ok else 0, 1 if gather_ok in (None, True) else 0], device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok, gather_ok = bool(okt[0].item()), (None if args.gather == "none" else bool(okt[1].item()))

    gather_ran = ss.gather_ran() if world > 1 else "none"           # the mode the library EXECUTED (p25fe_shard_gather_ran)
    rccl_evidence = shard_evidence(ss, dist, world) if world > 1 else None
    strong_extra, strong_ok = None, True
    if world > 1 and not strong and not args.no_extra:
        # configs[4] AS WORDED rides along as an extra row of the default (weak) line: one 3 600 s capture cut N ways
        try:
            stage("the strong row (configs[4] as worded): shard create + steps")
            strong_extra, strong_ok = strong_row(torch, dist, args, world, rank, local, dev, staged, max(3, min(args.steps, 100)),
                                                 args.strong_seconds)
        except Exception as e:
            strong_extra, strong_ok = {"config": "configs[4] as worded (strong)", "error": "%s: %s" % (type(e).__name__, e)}, False
    if rank == 0:
        total_samples = float(n) * world * args.steps
        value = total_samples / dt / 1e6
        k1_ms = kms[0] / max(ncalls, 1)
        achieved = BYTES_PER_SAMPLE * n / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        achieved_k1_alone = BYTES_PER_SAMPLE_K1 * n / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        traffic, traffic_source, traffic_warning = None, None, None
        if world == 1 and abs(total_s - 600.0) < 1e-9:
            for pmc_file in PMC_FILES:
                try:
                    with open(os.path.join(ROOT, pmc_file)) as f:
                        traffic = json.load(f).get("hbm_bytes_per_launch")
                    traffic_source = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel on this workload, collected in "
                                      "a SEPARATE run (tools/prof.sh), not by this process" % pmc_file)
                    # the counters were read from SOME build of K1: say so when it was not the source this run's library was built from
                    import hashlib
                    with open(os.path.join(ROOT, "p25rx_amd", "csrc", "p25fe_kernels.hip"), "rb") as kf:
                        sha = hashlib.sha256(kf.read()).hexdigest()[:16]
                    with open(os.path.join(ROOT, pmc_file)) as f:
                        was = json.load(f).get("k1_source_sha16")
                    if was != sha:
                        traffic_warning = ("STALE: %s was collected on %s, this run's p25fe_kernels.hip is %s -- `traffic` describes another build "
                                           "of K1; a traffic regression since then is NOT visible in this line (re-run tools/prof.sh)"
                                           % (pmc_file, ("K1 source " + was) if was else "an unrecorded K1 source", sha))
                    break
                except Exception:
                    traffic = None
        if world == 1:
            workload = ("configs[1]: 1 channel x %.0f s synthetic C4FM cf32 IQ @ 240 ksps (%d samples, %.3f GB), decimating "
                        "FIR + FM + boxcar + sync + 4-level slice" % (total_s, n, n * 8 / 1e9))
            sharding = "none"
        elif strong:
            workload = ("configs[4]: ONE %.0f s capture (%d samples, %.3f GB) cut into %d contiguous time shards of %.1f s, "
                        "decimating FIR + FM + boxcar + sync + 4-level slice + dibit gather (%s)"
                        % (total_s, n * world, n * world * 8 / 1e9, world, n / 240000.0, gather_ran))
            sharding = "time shards (strong), halo %d samples by send/recv behind K1, summaries by all_gather, dibits by %s (%s)" % (
                halo, GATHER_WORDS[gather_ran],
                "TEST HOOK: shared-memory staging, all ranks on one GPU" if staged else "RCCL inside libp25fe_rccl.so")
        else:
            workload = ("configs[4]'s partitioning at configs[1]'s per-GPU size: ONE %.0f s capture (%d samples, %.3f GB) cut into %d "
                        "contiguous time shards of %.0f s, decimating FIR + FM + boxcar + sync + 4-level slice + dibit gather (%s)"
                        % (total_s * world, n * world, n * world * 8 / 1e9, world, total_s, gather_ran))
            sharding = "time shards (weak), halo %d samples by send/recv behind K1, summaries by all_gather, dibits by %s (%s)" % (
                halo, GATHER_WORDS[gather_ran],
                "TEST HOOK: shared-memory staging, all ranks on one GPU" if staged else "RCCL inside libp25fe_rccl.so")
        out = {
            "metric": "IQ Msamples/s through FM-demod+C4FM slice",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "sharding": sharding,
                       "parity_gate": "dibits == modulator symbols: %s" % ok},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_warning": traffic_warning,
                         "traffic_ratio": (round(traffic / (BYTES_PER_SAMPLE * n), 4) if traffic else None),
                         "kernel": "k_frontend<cf32>", "kernel_ms": round(k1_ms, 4), "kernel_ms_samples": int(ncalls),
                         "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE,
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n,
                         "k1_alone": {"bytes_per_sample": BYTES_PER_SAMPLE_K1, "achieved": round(achieved_k1_alone, 1),
                                      "frac": round(achieved_k1_alone / HBM_PEAK_GBPS, 4),
                                      "note": "K1 priced on what it reads AND writes (8 B in + 0.8 B polyphase baseband out); "
                                              "the baseband is an intermediate of this design, so `frac` above uses SURVEY 8(d)'s 8.02 B"},
                         "vs_library_streams": ({"torch_copy_GBps": round(copy_gbps, 1), "torch_sum_read_only_GBps": round(sum_gbps, 1),
                                      "k1_traffic_GBps": (round(traffic / (k1_ms * 1e-3) / 1e9, 1) if (traffic and k1_ms > 0) else None),
                                      "k1_traffic_over_copy": (round(traffic / (k1_ms * 1e-3) / 1e9 / copy_gbps, 4) if (traffic and k1_ms > 0) else None),
                                      "note": "reference points measured by this process on the same capture, after the timed region: "
                                              "torch's device-to-device copy (read + write bytes / time) and torch's sum (read-only); "
                                              "k1_traffic_GBps = the PMC traffic above / this run's K1 time -- the HBM stream K1 sustains "
                                              "WHILE doing the arithmetic of SPEC section 3"} if copy_gbps else None),
                         "whole_step": {"algorithmic_bytes_per_sample": 8.02,
                                        "achieved": round(8.02 * n * world * args.steps / dt / 1e9, 1),
                                        "frac_of_peak_x_gpus": round(8.02 * n * args.steps / dt / 1e9 / HBM_PEAK_GBPS, 4)},
                         "other_kernels_ms": {"k_detect": round(kms_all[1] / n_extra_steps, 4),
                                              "k_scan_tiles": round(kms_all[2] / n_extra_steps, 4),
                                              "k_slice": round(kms_all[3] / n_extra_steps, 4),
                                              "note": "from extra SERIAL steps after the timed region, HIP events around every kernel"
                                                      + ("; N > 1: RCCL time is in none of them" if world > 1 else "")}},
        }
        if world == 1:
            out["config"]["prewarm"] = "%d untimed steps (>= %.0f ms) before the W warm-up steps" % (n_prewarm, PREWARM_MS)
            # for tools/prof_summary.py: which k_frontend dispatches of this process (0-based, in launch order) are the timed ones,
            # so that a rocprofv3 kernel trace of this command is summarised over the same launches `roofline.kernel_ms` covers
            out["config"]["k1_launch_range"] = [n_prewarm + args.warmup, args.steps]
            out["config"]["step"] = ("serial: K1 -> K2 -> K3 -> K4 on one stream" if args.no_pipeline else
                                     "pipelined two deep (p25fe_run_dev_pipelined): K2-K4 of step i on the handle's stream overlap "
                                     "K1 of step i + 1; join + device synchronize inside the timed region")
            if serial_ms is not None:
                out["config"]["serial_ms_per_step"] = round(serial_ms, 4)
        if world > 1:
            out["config"]["step"] = ("p25fe_shard_step_pipelined: K1's main launch on the caller's stream, halo + head on the side stream, "
                                     "detection / scan / all-gather / slicer / gather / compaction on the handle's receive stream beside "
                                     "the next step's K1; join + device synchronize inside the timed region" if shard_pipe else
                                     "p25fe_shard_step: every launch of a step behind its K1 on one stream")
        if comm_ms is not None:
            out["config"]["comm_ms_per_step"] = dict(comm_ms, note="rank 0, p25fe_shard_comm_ms: HIP events of the library around each "
                                                     "exchange on its stream (the halo exchange runs beside K1's main launch), on every "
                                                     "16th step of the timed region only (p25fe_shard_comm_timing: four of these "
                                                     "events are packets between the kernels of the step's critical path)")
        if rccl_evidence is not None:
            out["config"]["rccl"] = rccl_evidence
        if strong_extra is not None:
            out["extra"] = [strong_extra]
        if gather_ok is not None:
            out["config"]["gather_gate"] = ("gathered stream holds every shard at its resolved offset (length, sum and "
                                            "position-weighted sum of every rank's dibits): %s" % gather_ok)
        if world == 1 and not args.no_cpu:
            ncpu = min(n, int(args.cpu_seconds * 240000))
            host = iq[:ncpu].cpu().numpy().view(np.complex64).reshape(-1)
            cb, cpu_dib = cpu_baseline(host, "first %.0f s" % (ncpu / 240000))
            out["cpu_baseline"] = cb
            kk = min(len(cpu_dib), nd)
            # the oracle on the same samples must agree with the GPU bit for bit (full prefix)
            out["config"]["oracle_gate"] = "GPU dibits == oracle dibits on the CPU sample: %s" % bool(
                np.array_equal(cpu_dib[:kk - 1], got[:kk - 1]))
        if world == 1 and not args.no_extra:
            try:
                out["extra"] = run_extras(torch, dev, args, iq, truth)
            except Exception as e:                         # the headline line must not be lost to an extra
                out["extra_error"] = "%s: %s" % (type(e).__name__, e)
        print(json.dumps(out))
    if dist:
        ss.close()
        dist.barrier()
        if staged and rank == 0:
            try:
                os.unlink("/dev/shm" + boot[0])
            except OSError:
                pass
        dist.destroy_process_group()
    if not ok or gather_ok is False or not strong_ok:
        sys.exit(3)                        # a wrong result is a failed bench at every N (the gates are in the JSON line as well)


if __name__ == "__main__":
    main()
