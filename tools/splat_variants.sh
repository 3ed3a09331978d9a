#!/bin/bash
# A/B of k_splat_scatter builds on the light tracer's leg.  Usage (inside gpurun): bash tools/splat_variants.sh <lib> [<lib> ...]
for lib in "$@"; do
  for rep in 1 2; do
  MI355_LIB=$PWD/$lib timeout -k 10 200 python bench.py --only ptracer --no-ab > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1]); p = d["ptracer"]
print(sys.argv[1], "box", p["box"]["kernel_ms"], "gaussian", p["gaussian"]["kernel_ms"], "bwd all", p["gaussian"].get("render_backward_all_params_ms"), "fwd", p["gaussian"].get("render_forward_ms"), flush=True)
PY
  done
done
