#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of a build: python tools/kernel_regs.py [extra hipcc flags]  (compiles with
-Rpass-analysis=kernel-resource-usage into /tmp; no GPU needed)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log = subprocess.run([sys.executable, os.path.join(ROOT, "mitsuba3-camera_amd", "build.py"), "--out=/tmp/kernel_regs.so", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                     capture_output=True, text=True)
t = log.stdout + log.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip()
    g = lambda k: (lambda m: int(m.group(1)) if m else -1)(re.search(k + r": (\d+)", b))
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(.*", "", name.replace("void ", ""))
    print("%-48s vgpr %3d agpr %3d spilled %3d scratch %4d B lds %6d B  waves/SIMD %d" % (name[:48], g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
