"""Build librt355.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m raytracer_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
tree (it is not in .gpurunignore).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "rt355.hip")
HDR = os.path.join(os.path.dirname(HERE), "include", "rt355.h")
SO = os.path.join(CSRC, "librt355.so")

# -amdgpu-atomic-optimizer-strategy=None: the stream kernel keeps a lane-0 returning atomic (its next
# work claim) in flight across a whole shading pass; LLVM's atomic optimizer would rewrite it into a
# wave-aggregated form that reads the result back (s_waitcnt vmcnt(0)) right after issuing it.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-gpu-sanitize", "-Wall", "-Wno-unused-function",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
               "-munsafe-fp-atomics"]   # hardware float atomics for the refraction path's pixel sums


def sources() -> list[str]:
    """rt355.hip and the parts it includes (one translation unit), plus the public header."""
    return [SRC, HDR] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))


def needs_build() -> bool:
    if not os.path.isfile(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(f) > t for f in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, "-o", SO, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
