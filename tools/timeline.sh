#!/bin/bash
# Kernel timeline of the lens leg: where the time of one render / one adjoint call goes BETWEEN its kernels.
# Usage (inside gpurun): bash tools/timeline.sh <tag>   -> gpurun_out/timeline_<tag>.txt
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/bench.py --steps 2 --warmup 1 --only lens --no-ab --no-cpu > $OUT/bench.json 2> $OUT/bench.err
echo "exit $?"
python3 $R/tools/timeline.py $OUT/raw > $R/gpurun_out/timeline_$TAG.txt
rm -rf $OUT/raw
tail -60 $R/gpurun_out/timeline_$TAG.txt
