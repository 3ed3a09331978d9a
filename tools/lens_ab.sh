#!/bin/bash
# A/B of builds on the lens leg of bench.py (BASELINE configs[4]).  Usage (inside gpurun): bash tools/lens_ab.sh <lib> [<lib> ...]
# A library that cannot hold the doublet's BVH fails with the engine's build-time message.
for lib in "$@"; do
  MI355_LIB=$lib python bench.py --only lens --steps 2 --warmup 1 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json, sys
l = json.load(open("/tmp/ab.json"))["lens"]; s = l["sensors"][0]
print(sys.argv[1], "render", s["render_256spp_ms"], "(kernels", s["render_kernels_ms"], ") geometry adjoint", s["render_backward_geometry_ms"], "(kernels", s["render_backward_kernels_ms"], ") iteration", l["design_iteration_ms"])
PY
done
