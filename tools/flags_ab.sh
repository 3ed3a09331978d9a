#!/bin/bash
# A/B of whole-library builds (compiler flags) on every leg.  Usage (inside gpurun): bash tools/flags_ab.sh <lib> [<lib> ...]
for lib in "$@"; do
  MI355_LIB=$PWD/$lib timeout -k 10 300 python bench.py --only prb,veach,ptracer,lens --steps 10 --warmup 3 --no-ab > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -n 3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1]); L = d["lens"]
print(sys.argv[1], "C2", d["roofline"]["kernel_ms"], "| prb", d["prb"]["kernel_ms"], "| veach", d["veach"]["kernel_ms"], "| ptracer", d["ptracer"]["box"]["kernel_ms"], d["ptracer"]["gaussian"]["kernel_ms"],
      "| lens render/adjoint one pass", L.get("render_all_sensors_one_pass_ms"), L.get("render_backward_all_sensors_one_pass_ms"), "iteration", L.get("design_iteration_ms"), flush=True)
PY
done
