#!/bin/bash
# rocprofv3 kernel statistics (+ roctx marker ranges) of bench.py legs.  Usage (inside gpurun): bash tools/kstats.sh <tag> <legs> [bench args]
#   e.g. bash tools/kstats.sh lens lens      -> gpurun_out/kstats_lens/{kernel_stats.csv,marker_stats.csv,bench.json}
TAG=${1:-run}; LEGS=${2:-prb}; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/raw -- python3 $R/bench.py --steps 4 --warmup 1 --only $LEGS $* > $OUT/bench.json 2> $OUT/bench.err
echo "exit $?"
cp $OUT/raw/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cp $OUT/raw/*/*_marker_api_stats.csv $OUT/marker_stats.csv 2>/dev/null || cp $OUT/raw/*/*marker*stats*.csv $OUT/marker_stats.csv 2>/dev/null
rm -rf $OUT/raw
head -n 14 $OUT/kernel_stats.csv | cut -c1-160
