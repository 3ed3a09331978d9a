#!/bin/bash
# rocprofv3 kernel statistics of any python script of this repository.  Usage (inside gpurun): bash tools/kstats_py.sh <tag> <script> [args]
TAG=${1:-run}; SCRIPT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $R/$SCRIPT $* > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "exit $?"
cp $OUT/raw/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/raw
head -n 40 $OUT/kernel_stats.csv | cut -c1-200
tail -12 $OUT/stdout.txt
