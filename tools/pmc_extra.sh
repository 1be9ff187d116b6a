#!/bin/bash
# Extra PMC pass for K1 latency analysis (never combined with tracing).  usage: tools/pmc_extra.sh ; output gpurun_out/pmc_extra/
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_extra; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu > $OUT/pmc_$i.log 2>&1
done <<'GROUPS'
SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA
SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU
GROUPS
cd $ROOT
python3 tools/prof_summary.py $OUT 2>/dev/null | grep "k_frontend\|k_sync" 
