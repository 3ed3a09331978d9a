#!/bin/bash
# Memory-pipeline counters (texture addresser / L1 / instruction cache / LDS) of one bench.py leg.  Usage (inside gpurun): LEGS=lens bash tools/pmc_mem.sh <tag>
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-ab --only ${LEGS:-lens} $*"
i=0
for P in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
         "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
         "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass$i exit $?"
done
python3 $R/tools/pmc_summary.py $OUT $OUT/summary.txt > /dev/null 2>&1
echo "summary exit $?"
rm -rf $OUT/pass[0-9]* $OUT/stats
