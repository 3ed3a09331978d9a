#!/bin/bash
# A/B of builds on the headline leg (BASELINE configs[1], forward kernel only).  Usage (inside gpurun): bash tools/c2_ab.sh <lib> [<lib> ...]
for lib in "$@"; do
  MI355_LIB=$lib python bench.py --only none --steps 10 --warmup 3 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "Msamples/s", d["value"])
PY
done
