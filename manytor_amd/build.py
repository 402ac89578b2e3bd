"""Build libmanytor_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m manytor_amd.build            # build if sources are newer than the .so
    python -m manytor_amd.build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libmanytor_hip.so")
SOURCES = ["engine.hip", "comm.hip"]
DEPS = ["engine.hip", "comm.hip", "engine_internal.h", "step_args.h", "kernels.h", "mt_math.h", "philox.h"]


EXTRA_FLAGS = [
    # let x*0 and x+0 fold (static DH tables); staged actions are screened on their bit pattern (unusable_angle), so
    # no NaN/inf reaches the arithmetic of the step path
    "-fno-signed-zeros", "-ffinite-math-only",
    # gfx950 issues v_pk_*_f32 at half the rate of scalar VALU ops, so SLP-packing adjacent fp32 math only adds
    # register shuffles (measured: sub-step loop 66 -> 43 instructions, 72 -> 58 VGPRs without it)
    "-fno-slp-vectorize",
    # FMAs are written out in the source; no implicit contraction, so every kernel rounds a formula identically
    "-ffp-contract=off",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(INCLUDE, "manytor_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [
        _hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden",
        "-I", INCLUDE, "-Wall", "-Wno-unused-function",
        *EXTRA_FLAGS,
        *extra_flags,
        *[os.path.join(CSRC, s) for s in SOURCES],
        "-o", LIB_PATH + ".tmp",
    ]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
