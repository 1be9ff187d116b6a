#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel average duration and PMC counters per launch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
def short(k):
    for s in ("k_frontend", "k_sync", "k_scan", "k_slice", "k_power"):
        if s in k:
            return s
    return None
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --kernel-trace --stats:", os.path.relpath(f, d))
    for r in csv.DictReader(open(f)):
        if short(r.get("Name", "")):
            print("  %-12s calls %4s  avg %10.1f ns  min %9s  max %9s  pct %s" % (
                short(r["Name"]), r.get("Calls"), float(r.get("AverageNs", 0)), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r.get("Kernel_Name", ""))
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== PMC:", os.path.relpath(f, d))
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            print("  %-12s %-26s per-launch mean %16.1f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
