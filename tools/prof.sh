#!/bin/bash
# rocprofv3 recipe (run on the GPU box via gpurun): kernel-trace stats of bench.py + PMC passes (separate runs,
# --pmc never combined with tracing).  usage: tools/prof.sh <tag> ; outputs under gpurun_out/prof_<tag>/
TAG=${1:-x}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/stats.log 2>&1
grep -o '{"metric.*' $OUT/stats.log > $OUT/bench_under_rocprof.json
# the same with strictly serial steps: kernel durations without the overlap of the pipelined form
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --no-pipeline > $OUT/stats_serial.log 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-extra > $OUT/pmc_$i.log 2>&1
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
cd $ROOT
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
