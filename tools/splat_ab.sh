#!/bin/bash
# Parity tests that go through the deferred splat, then the light tracer's bench leg.  Usage (inside gpurun): bash tools/splat_ab.sh <tag>
TAG=${1:-run}
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "particle or ptracer or splat or tracer or sensor or fields or geometry" > gpurun_out/splat_tests_$TAG.log 2>&1
tail -3 gpurun_out/splat_tests_$TAG.log
timeout -k 10 300 python bench.py --only ptracer > gpurun_out/splat_bench_$TAG.json 2> gpurun_out/splat_bench_$TAG.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/splat_bench_$TAG.json").read().strip().splitlines()[-1]); p = d["ptracer"]
print({k: (v.get("kernel_ms") if isinstance(v, dict) else v) for k, v in p.items()})
PY
