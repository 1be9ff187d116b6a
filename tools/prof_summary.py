#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory.

* K1 of the bench: duration over the TIMED launches only.  bench.py prints which k_frontend dispatches of its process are the
  timed ones (config.k1_launch_range); the rocprofv3 kernel trace of the same command is cut to exactly those, so that this
  summary and the line's roofline.kernel_ms cover the same launches (the aggregate --stats average also holds the cold
  pre-warm launches and the serial extra steps: round 3's 268 us against the line's 246).
* every other p25k kernel: calls / median / mean / p10 / p90 from the trace, with the workload size stated by the caller.
* PMC passes: per-launch means, and K1's HBM traffic with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section:
  FETCH_SIZE reports half the bytes of a wide coalesced read; both counters are in KiB)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

d = sys.argv[1]


def short(k):
    m = re.search(r"p25k::(\w+)(<[^(]*>)?", k)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r"(p25jit_\w+)", k)
    return m.group(1) if m else None


def trace_rows(sub):
    rows = []
    for f in glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if k:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), k))
    rows.sort()
    return rows


def stats(v):
    v = sorted(v)
    n = len(v)
    return dict(calls=n, mean=sum(v) / n, median=v[n // 2], p10=v[n // 10], p90=v[min(n - 1, (9 * n) // 10)])


def bench_json(name):
    try:
        txt = open(os.path.join(d, name)).read()
        return json.loads(txt[txt.index('{"metric'):].splitlines()[0])
    except Exception:
        return None


for sub, log, what in (("stats", "stats.log", "pipelined steps (the bench's default)"), ("stats_serial", "stats_serial.log", "--no-pipeline (every kernel alone on the chip)")):
    rows = trace_rows(sub)
    if not rows:
        continue
    b = bench_json(log)
    print("== rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extra [%s]" % what)
    k1 = [dur for _, dur, k in rows if k.startswith("k_frontend")]
    if b and "k1_launch_range" in b.get("config", {}):
        first, cnt = b["config"]["k1_launch_range"]
        timed = k1[first:first + cnt]
        st = stats(timed)
        line_ms = b["roofline"]["kernel_ms"]
        print("  k_frontend, the %d TIMED launches (dispatches %d .. %d of %d): mean %.1f us  median %.1f  p10 %.1f  p90 %.1f"
              % (cnt, first, first + cnt - 1, len(k1), st["mean"] / 1e3, st["median"] / 1e3, st["p10"] / 1e3, st["p90"] / 1e3))
        print("  the same run's JSON line: roofline.kernel_ms %.4f (HIP events of the library on the launch stream) -> trace / line = %.4f"
              % (line_ms, st["mean"] / 1e6 / line_ms))
        json.dump({"k1_timed_mean_us": st["mean"] / 1e3, "k1_timed_median_us": st["median"] / 1e3, "line_kernel_ms": line_ms,
                   "ratio": st["mean"] / 1e6 / line_ms, "launches": cnt}, open(os.path.join(d, "k1_trace_vs_line_%s.json" % sub), "w"))
    by = collections.defaultdict(list)
    for _, dur, k in rows:
        by[k].append(dur)
    for k in sorted(by):
        st = stats(by[k])
        print("  %-44s calls %5d  median %9.1f us  p10 %9.1f  p90 %9.1f  mean %9.1f   (all launches of the process)"
              % (k[:44], st["calls"], st["median"] / 1e3, st["p10"] / 1e3, st["p90"] / 1e3, st["mean"] / 1e3))

for sub in sorted(glob.glob(os.path.join(d, "kb_*"))):
    rows = trace_rows(os.path.basename(sub))
    if not rows:
        continue
    note = ""
    try:
        note = open(os.path.join(sub, "WORKLOAD")).read().strip()
    except OSError:
        pass
    print("== rocprofv3 --kernel-trace -- python3 tools/kbench.py : %s" % note)
    timed = None                                            # kbench's pre-warm launches are part of the trace: keep the timed rounds
    try:
        m = re.search(r"kbench timed_rounds=(\d+)", open(sub + ".log").read())
        timed = int(m.group(1)) if m else None
    except OSError:
        pass
    by = collections.defaultdict(list)
    for _, dur, k in rows:
        by[k].append(dur)
    if timed:
        print("  (the last %d launches of every kernel: the timed rounds behind kbench's >= 150 ms pre-warm)" % timed)
    for k in sorted(by):
        v = by[k][-timed:] if timed else by[k]
        st = stats(v)
        print("  %-44s calls %5d  median %9.1f us  p10 %9.1f  p90 %9.1f  mean %9.1f"
              % (k[:44], st["calls"], st["median"] / 1e3, st["p10"] / 1e3, st["p90"] / 1e3, st["mean"] / 1e3))

vals = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r.get("Kernel_Name", ""))
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== PMC (separate --pmc run):", os.path.relpath(f, d))
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            vals[k][c] = sum(v) / len(v)
            print("  %-44s %-26s per-launch mean %16.1f  (n=%d)" % (k[:44], c, sum(v) / len(v), len(v)))
for k, tag in ((next((x for x in vals if x.startswith("k_frontend<0")), None), "cf32"), (next((x for x in vals if x.startswith("k_frontend<1")), None), "u8")):
    if not k:
        continue
    k1 = vals[k]
    if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
        rd = 2.0 * k1["FETCH_SIZE"] * 1024.0          # gfx950: FETCH_SIZE counts 64 B per 128-B request -> double it
        wr = k1["WRITE_SIZE"] * 1024.0
        out = {"kernel": "k_frontend<%s>" % tag, "fetch_size_kib_raw": k1["FETCH_SIZE"], "write_size_kib_raw": k1["WRITE_SIZE"],
               "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE as reported; both KiB",
               # which K1 the counters were read from: bench.py compares it with the source it runs and says so when they differ
               "k1_source_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "p25rx_amd", "csrc",
                                                                   "p25fe_kernels.hip"), "rb").read()).hexdigest()[:16]}
        print("== K1<%s> HBM traffic per launch:" % tag, json.dumps(out))
        json.dump(out, open(os.path.join(d, "k1_pmc%s.json" % ("" if tag == "cf32" else "_u8")), "w"), indent=1)
