"""Builds libmi355.so (HIP, gfx950 only) in-tree, next to this file.

hipcc cross-compiles without a GPU, so this runs in the CPU-only authoring container as
well as on the MI355X box.  The library is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "mi355_engine.hip")]
HDR = [os.path.join(HERE, "csrc", h) for h in ("mi355_device.h", "mi355_geo_forward.h", "mi355_geo_adjoint.h", "mi355_geo_backward.h")] + [os.path.join(ROOT, "include", "mi355.h")]
LIB = os.path.join(HERE, "libmi355.so")

# -ffp-contract=off: only the fmaf() written in the source fuse (same rule as the CPU oracle and as
# Dr.Jit's explicit fmadd), so lanes agree with the reference arithmetic instead of with the optimiser.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-ldl",
]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=False, extra_flags=(), out=None):
    if out is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-o", out or LIB] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose and r.stderr:
        print(r.stderr)
    return LIB


if __name__ == "__main__":
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    build(force="--force" in sys.argv, verbose=True, out=outs[0] if outs else None,
          extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force" and not a.startswith("--out=")])
    print(outs[0] if outs else LIB)
