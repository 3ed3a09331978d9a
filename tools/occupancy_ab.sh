#!/bin/bash
# A/B of resident waves per SIMD for the forward kernel (BASELINE configs[1]): the same code (register budget of 5 waves), the
# occupancy lowered through extra LDS per workgroup.  Libraries are built beforehand (CPU container):
#   python mitsuba3-camera_amd/build.py --out=build/libmi355_w5.so -DMI_MIN_WAVES=5
#   python mitsuba3-camera_amd/build.py --out=build/libmi355_w4.so -DMI_MIN_WAVES=5 -DMI_LDS_PAD=2048
#   python mitsuba3-camera_amd/build.py --out=build/libmi355_w3.so -DMI_MIN_WAVES=5 -DMI_LDS_PAD=4096
#   python mitsuba3-camera_amd/build.py --out=build/libmi355_prof.so -DMI_MIN_WAVES=5 -DMI_PROF      (s_memtime sections)
# Usage (inside gpurun): bash tools/occupancy_ab.sh > gpurun_out/occupancy_ab.txt
for v in w5 w4 w3 prof; do
  MI355_LIB=build/libmi355_$v.so python bench.py --only none --steps 6 --warmup 2 > /tmp/ab_$v.json 2> /tmp/ab_$v.err || { echo "$v failed"; cat /tmp/ab_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]; d = json.load(open("/tmp/ab_%s.json" % v))
print(v, "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
  grep "MI_PROF" /tmp/ab_$v.err | tail -n 1
done
