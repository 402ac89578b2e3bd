#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of libmanytor_hip.so's device code (hipcc
-Rpass-analysis=kernel-resource-usage), demangled.  python tools/kernel_resources.py [filter ...] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manytor_amd import build as B  # noqa: E402


def main():
    flt = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        cmd = [B._hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", "-I", B.INCLUDE, *B.EXTRA_FLAGS,
               "-Rpass-analysis=kernel-resource-usage", os.path.join(B.CSRC, "engine.hip"), "-o", os.path.join(td, "e.o")]
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'SGPR':>5} {'scratch':>7} {'LDS':>6} {'waves/SIMD':>10}  kernel")
    for b, nm in zip(blocks, dem):
        def g(k):
            mm = re.search(k + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1
        nm = re.sub(r"^void ", "", nm)
        if flt and not any(f in nm for f in flt):
            continue
        scratch, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{g('VGPRs'):>5} {g('SGPRs'):>5} {scratch:>7} {lds:>6} {occ:>10}  {nm[:150]}")


if __name__ == "__main__":
    main()
