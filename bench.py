#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through FM-demod + C4FM slice on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (K1 front end -> K2 sync -> K3 scan -> K4 slice) over one
resident capture: BASELINE.json configs[1], 1 channel x 600 s of synthetic C4FM IQ at 240 ksps
(1.44e8 cf32 samples, 1.152 GB) per GPU, generated in HBM before the timed region.

N > 1 (one rank per GPU, RCCL): BASELINE.json configs[4]'s structure -- one long capture cut
into N contiguous 600 s time shards (weak scaling).  Each step exchanges the filter halo with
the left neighbour (send/recv), runs pass 1, all-gathers the 56-byte shard summaries, resolves
the symbol-timing carry on the device and runs pass 2 -- no host synchronisation inside a step.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
dominant kernel (K1) from HIP events recorded inside the library on the launch stream, and
`cpu_baseline` from the CPU oracle timed on this box's host cores (rank 0, N = 1 only).
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
BYTES_PER_SAMPLE_K1 = 8.0 + 0.8  # algorithmic: 8 B cf32 read + 4 B baseband written per 5 samples


def cpu_baseline(iq_host, seconds_label):
    """Time the oracle (scalar C port of the reference structure) on one host core."""
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
    # second figure (SURVEY 8d ii): the same port on all host cores, the capture cut into one time shard per thread
    # (pthreads inside the oracle; shard-boundary symbols are not stitched -- a throughput figure only)
    ncores = os.cpu_count() or 1
    O.run_cf32_mt(iq_host, ncores, libpath=so)                    # untimed: the first multi-threaded pass runs at one core's
    t0 = time.perf_counter()                                      # pace on these hosts (idle cores take a while to come up)
    O.run_cf32_mt(iq_host, ncores, libpath=so)
    dt_all = time.perf_counter() - t0
    if so and os.path.exists(so):
        os.unlink(so)
    return {"value": round(len(iq_host) / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%s of the same capture (%d IQ samples), oracle/p25fe_oracle.c built %s, 1 thread, %.2f s"
                      % (seconds_label, len(iq_host), kind_flags, dt),
            "all_cores": {"value": round(len(iq_host) / dt_all / 1e6, 3), "cores": ncores,
                          "note": "same sample as %d independent time shards, one thread each, %.2f s" % (ncores, dt_all)}}, dib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seconds", type=float, default=600.0, help="capture length per GPU (configs[1]: 600)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=600.0, help="length of the capture prefix timed on the CPU")
    args = ap.parse_args()

    import torch
    from p25rx_amd import c4fm
    from p25rx_amd.frontend import FrontEnd, parse_results, n_baseband
    from p25rx_amd._lib import RESULT_DTYPE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("P25FE_BENCH_ONE_GPU"):        # test hook: every rank on GPU 0 (only useful where RCCL allows it)
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = int(round(args.seconds * 240000))
    n -= n % 8                                                     # shard cut points stay 16-B aligned
    fe = FrontEnd(device=local)
    halo = fe.shard_halo()
    # resident capture: [halo | owned shard]; the halo region is filled by the neighbour exchange
    buf = torch.zeros((halo + n, 2), dtype=torch.float32, device=dev)
    iq = buf[halo:]
    _, truth = c4fm.synth_torch(n, seed=1000 + rank, device=dev, snr_db=30.0, out=iq)
    torch.cuda.synchronize()

    cap = (n // 50 + 64 + 15) // 16 * 16
    dibits = torch.empty((1, cap), dtype=torch.uint8, device=dev)
    result = torch.empty((1, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    abs0 = rank * n
    bb0 = n_baseband(0, abs0)
    bbn = n_baseband(abs0, n)

    if world == 1:
        def step():
            fe.run_dev(iq, dibits=dibits, result=result)
    else:
        from p25rx_amd.sharding import TimeShard
        ts = TimeShard(fe, rank, world, n, dist)
        ts.setup_device(torch, dev)
        summ_all = torch.empty((world, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)

        def step():
            ts.step_device(buf, result, summ_all, dibits)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fe.profile_enable(2)            # HIP events around K1 only inside the timed region (each record costs ~3 us of gap)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, ncalls = fe.profile_read()
    # per-kernel split of the other kernels: a few extra steps OUTSIDE the timed region with events around every kernel
    fe.profile_enable(1)
    for _ in range(min(args.steps, 8)):
        step()
    torch.cuda.synchronize()
    kms_all, ncalls_all = fe.profile_read()
    fe.profile_enable(False)
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # correctness gate on what the timed steps produced: demod(mod(d)) == d for this rank's shard
    r = parse_results(result)[0]
    nd = int(r["n_dibits"])
    got = dibits[0, :nd].cpu().numpy()
    ok = True
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
    if dist:
        okt = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        total_samples = float(n) * world * args.steps
        value = total_samples / dt / 1e6
        k1_ms = kms[0] / max(ncalls, 1)
        achieved = BYTES_PER_SAMPLE_K1 * n / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_k1_pmc.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "IQ Msamples/s through FM-demod+C4FM slice",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 1 channel x %.0f s synthetic C4FM cf32 IQ @ 240 ksps per GPU "
                                   "(%d samples, %.3f GB), decimating FIR + FM + boxcar + sync + 4-level slice"
                                   % (args.seconds, n, n * 8 / 1e9),
                       "sharding": "none" if world == 1 else "time shards, halo %d samples, RCCL send/recv + all_gather" % halo,
                       "parity_gate": "dibits == modulator symbols: %s" % ok},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "k_frontend<cf32>", "kernel_ms": round(k1_ms, 4),
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE_K1 * n,
                         "other_kernels_ms": {"k_sync": round(kms_all[1] / max(ncalls_all, 1), 4),
                                              "k_scan": round(kms_all[2] / max(ncalls_all, 1), 4),
                                              "k_slice": round(kms_all[3] / max(ncalls_all, 1), 4),
                                              "note": "from extra steps after the timed region"}},
        }
        if world == 1 and not args.no_cpu:
            ncpu = min(n, int(args.cpu_seconds * 240000))
            host = iq[:ncpu].cpu().numpy().view(np.complex64).reshape(-1)
            cb, cpu_dib = cpu_baseline(host, "first %.0f s" % (ncpu / 240000))
            out["cpu_baseline"] = cb
            kk = min(len(cpu_dib), nd)
            # the oracle on the same samples must agree with the GPU bit for bit (full prefix)
            out["config"]["oracle_gate"] = "GPU dibits == oracle dibits on the CPU sample: %s" % bool(
                np.array_equal(cpu_dib[:kk - 1], got[:kk - 1]))
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()
    if not ok and world == 1:
        sys.exit(3)                        # N = 1: a wrong result is a failed bench (N > 1 reports the gate in the JSON)


if __name__ == "__main__":
    main()
