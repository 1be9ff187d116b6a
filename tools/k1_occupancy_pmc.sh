#!/bin/bash
# The occupancy sweep of tools/k1_occupancy.sh under `rocprofv3 --pmc GRBM_GUI_ACTIVE`: GPU cycles AND duration of every K1 launch,
# i.e. the shader clock at every occupancy (the chip is power-limited: the clock follows the load).  One table on stdout.
ROOT=$PWD
OUT=$ROOT/gpurun_out/k1occ_pmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pad in 42924 28924 19924 10924 5424 1924 0; do
  for tag in cur abl1 abl6; do
    lib=$ROOT/build/variants/libp25fe_$tag.so; [ "$tag" == "cur" ] && lib=$ROOT/p25rx_amd/libp25fe.so
    [ -f $lib ] || continue
    export P25FE_K1_LDS_PAD=$pad P25FE_LIB=$lib
    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${tag}_$pad -o pmc -- python3 $ROOT/tools/k1_occupancy.py $tag 600 10 > $OUT/${tag}_$pad.log 2>&1
  done
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, statistics
print("%-6s %8s %3s %-5s %10s %10s %8s" % ("lib", "pad", "k", "fmt", "us", "cycles", "GHz"))
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*_*")), key=lambda p: (-int(p.rsplit("_", 1)[1]) if os.path.isdir(p) else 0, p)):
    if not os.path.isdir(d):
        continue
    tag, pad = os.path.basename(d).rsplit("_", 1)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_frontend" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    for fmt, key in (("cf32", "k_frontend<0"), ("u8", "k_frontend<1")):
        rr = [r for r in rows if key in r["Kernel_Name"]][-8:]          # the last launches: steady state
        if not rr:
            continue
        us = statistics.median((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rr)
        cyc = statistics.median(float(r["Counter_Value"]) / 8.0 for r in rr)
        print("%-6s %8s %3d %-5s %10.1f %10.0f %8.3f" % (tag, pad, 163840 // (13076 + int(pad)), fmt, us, cyc, cyc / us / 1e3))
PY
