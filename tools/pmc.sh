#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, never combined with tracing) over tools/kbench.py.
# usage: tools/pmc.sh <tag> <kbench args...>; prints per-kernel per-launch means.
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/tools/kbench.py "$@" > $OUT/p$i.log 2>&1
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        nm = r.get("Kernel_Name", "")
        if "p25k::" not in nm:
            continue
        m = re.search(r"p25k::(\w+)(<[^>]*>)?", nm)
        acc[m.group(1) + (m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
ks = sorted(acc)
cs = sorted({c for k in ks for c in acc[k]})
print("%-24s" % "counter" + "".join("%24s" % k[-24:] for k in ks))
for c in cs:
    print("%-24s" % c + "".join("%24.4g" % (sum(acc[k][c]) / max(1, len(acc[k][c]))) if c in acc[k] else "%24s" % "-" for k in ks))
PY
