#!/bin/bash
# rocprofv3 recipe (run on the GPU box via gpurun): kernel traces of bench.py (pipelined and serial) and of tools/kbench.py at
# the bench's own sizes (K0 / K6 / u8 / specialised tables), then PMC passes -- separate runs, --pmc never combined with tracing.
# usage: tools/prof.sh <tag> ; outputs under gpurun_out/prof_<tag>/ ; summary: tools/prof_summary.py (copy it to profiles/).
TAG=${1:-x}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/stats.log 2>&1
grep -o '{"metric.*' $OUT/stats.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --no-pipeline > $OUT/stats_serial.log 2>&1
# the other kernels at the bench's own sizes
mkdir -p $OUT/kb_u8 $OUT/kb_wide $OUT/kb_spec
echo "configs[1] as u8: 1 channel x 600 s = 1.44e8 samples, 20 rounds of run_u8 (K1<u8> planar -> K2 -> K3 -> K4) and lin_u8" > $OUT/kb_u8/WORKLOAD
rocprofv3 --kernel-trace --output-format csv -d $OUT/kb_u8 -o kb -- python3 $ROOT/tools/kbench.py 600 20 1 run_u8 lin_u8 > $OUT/kb_u8.log 2>&1
echo "configs[2] / channeliser at the bench's size: 1.44e8 wideband cf32 samples (60 s at 2.4 Msps), 8 rounds of k0 (k_predecim) and chz (k_channelise)" > $OUT/kb_wide/WORKLOAD
rocprofv3 --kernel-trace --output-format csv -d $OUT/kb_wide -o kb -- python3 $ROOT/tools/kbench.py 600 8 1 k0 chz > $OUT/kb_wide.log 2>&1
echo "configs[1] with non-default 31 / 41 tables: specialised (hipRTC) kernels p25jit_*, 20 rounds of run and run_u8" > $OUT/kb_spec/WORKLOAD
P25FE_KBENCH_TAPS=31,41 rocprofv3 --kernel-trace --output-format csv -d $OUT/kb_spec -o kb -- python3 $ROOT/tools/kbench.py 600 20 1 run run_u8 > $OUT/kb_spec.log 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $ROOT/tools/kbench.py 600 4 1 run run_u8 > $OUT/pmc_$i.log 2>&1
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
