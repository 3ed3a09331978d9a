#!/bin/bash
# A/B of builds on the headline, prb, veach and ptracer legs.  Usage (inside gpurun): bash tools/legs_ab.sh <lib> [<lib> ...]
for lib in "$@"; do
  MI355_LIB=$lib python bench.py --only prb,veach,ptracer --steps 10 --warmup 3 --no-ab > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "C2 kernel_ms", d["roofline"]["kernel_ms"], "| prb kernel_ms", d["prb"]["kernel_ms"], "render_forward_ms", d["prb"]["render_forward_ms"], "| veach ms", d["veach"]["ms_per_step"],
      "| ptracer box/gauss ms", d["ptracer"]["box"]["ms_per_step"], d["ptracer"]["gaussian"]["ms_per_step"], "bwd all", d["ptracer"]["gaussian"].get("render_backward_all_params_ms"))
PY
done
