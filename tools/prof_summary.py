#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel average duration and PMC counters per launch,
plus the HBM traffic of K1 per launch with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports half the bytes of a wide coalesced read; both counters are in KiB)."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
def short(k):
    if "p25k::" not in k:
        return None
    for s in ("k_frontend", "k_detect", "k_scan", "k_slice", "k_planarize", "k_power", "k_predecim", "k_channelise", "k_nid",
              "k_chan_stats", "k_shard"):
        if "p25k::" + s in k:
            return s
    return None
for sub, what in (("stats", "(pipelined steps, the bench's default: the receive kernels overlap the next K1 and wait for wave slots, "
                               "so their durations here are not their running times)"),
                  ("stats_serial", "--no-pipeline (serial steps: every kernel alone on the chip)")):
  for f in glob.glob(os.path.join(d, sub, "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extra", what, ":", os.path.relpath(f, d))
    for r in csv.DictReader(open(f)):
        if short(r.get("Name", "")):
            print("  %-12s calls %4s  avg %10.1f ns  min %9s  max %9s  pct %s" % (
                short(r["Name"]), r.get("Calls"), float(r.get("AverageNs", 0)), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
vals = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r.get("Kernel_Name", ""))
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== PMC (separate --pmc run of bench.py --steps 4 --no-extra):", os.path.relpath(f, d))
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            vals[k][c] = sum(v) / len(v)
            print("  %-12s %-26s per-launch mean %16.1f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
k1 = vals.get("k_frontend", {})
if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
    rd = 2.0 * k1["FETCH_SIZE"] * 1024.0          # gfx950: FETCH_SIZE counts 64 B per 128-B request -> double it
    wr = k1["WRITE_SIZE"] * 1024.0
    out = {"kernel": "k_frontend<cf32>", "fetch_size_kib_raw": k1["FETCH_SIZE"], "write_size_kib_raw": k1["WRITE_SIZE"],
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
           "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE as reported; both KiB"}
    print("== K1 HBM traffic per launch:", json.dumps(out))
    json.dump(out, open(os.path.join(d, "k1_pmc.json"), "w"), indent=1)
