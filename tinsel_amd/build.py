"""Builds tinsel_amd/libtinsel_hip.so (the C-ABI of include/tinsel_hip.h) with hipcc for gfx950.

The library is built IN-TREE so that it travels with the repository snapshot to the GPU box.
Flags that are part of the numerical contract (DESIGN.md "Arithmetic"):
  -ffp-contract=off   no FMA contraction: every fp32 op rounds once, like the CPU oracle's
  (no -ffast-math)    IEEE division / sqrt, ocml sinf/cosf/expf/logf/acosf/atan2f
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "tinsel_hip.hip")
OUT = os.path.join(HERE, "libtinsel_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))] + \
       [os.path.join(ROOT, "include", "tinsel_hip.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=True, extra=()):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra) + ["-o", OUT, SRC]
    if verbose:
        print("[tinsel_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
