#!/bin/bash
# Same-box kernel A/B under rocprofv3 --kernel-trace --stats.  usage: tools/ab.sh <tag> <kbench args...>
# (P25FE_LIB selects a measurement build).  Prints per-kernel average durations.
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ab -- python3 $ROOT/tools/kbench.py "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r.get("Kernel_Name", "")
        if "p25k::" not in nm:
            continue
        m = re.search(r"p25k::(\w+)(<[^>]*>)?", nm)
        d[m.group(1) + (m.group(2) or "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    v = v[3 * (len(v) > 12):]          # drop nothing; warm-up calls are simply part of the distribution tails
    print("  %-34s calls %4d  median %9.2f us  p10 %9.2f  p90 %9.2f  mean %9.2f" % (
        k, len(v), v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10], sum(v) / len(v)))
PY
