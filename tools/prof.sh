#!/bin/bash
# rocprofv3 recipe (run on the GPU box via gpurun): kernel-trace stats + PMC passes for the kernels.
# usage: tools/prof.sh <tag> [seconds] ; outputs under gpurun_out/prof_<tag>/
TAG=${1:-x}; SECS=${2:-600}
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $OLDPWD/tools/k1_bench.py $SECS 10 run > $OUT/stats.log 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp -d $OUT/pmc_$name -o pmc -- python3 $OLDPWD/tools/k1_bench.py $SECS 3 run > $OUT/pmc_$name.log 2>&1
done
cd $OLDPWD
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
