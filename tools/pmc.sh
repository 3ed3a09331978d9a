#!/bin/bash
# Collects rocprofv3 kernel-trace stats + PMC passes for bench.py's forward kernel on the GPU box.
# Usage (inside gpurun): bash tools/pmc.sh <tag> [extra bench args]
# Counters go in separate passes (SQ: 8 slots, TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass), never
# combined with sys/hip/hsa tracing.  Results: gpurun_out/pmc_<tag>/...; summarise with tools/pmc_summary.py.
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-ab --only ${LEGS:-prb} $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats exit $?"
i=0
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32" \
         "SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass$i exit $?"
done
# summarise on the box and drop the raw per-dispatch CSVs (hundreds of MB: gpurun_out/ is merged back only below 64 MiB)
python3 $R/tools/pmc_summary.py $OUT $OUT/summary.txt > /dev/null 2>&1
echo "summary exit $?"
rm -rf $OUT/pass[0-9]* $OUT/stats
