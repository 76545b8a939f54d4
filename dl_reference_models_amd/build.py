"""Builds libmapfstep.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m dl_reference_models_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so travels with the source tree (git-ignored).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libmapfstep.so")
SOURCES = [os.path.join(CSRC, "mapf_step.hip")]
DEVICE_INCLUDES = [os.path.join(CSRC, "mapf_kernels.inl")]
HEADERS = [os.path.join(ROOT, "include", "mapf_step.h")]

# NOTE: no -ffast-math -- goal_delta needs the correctly rounded fp32 divide.
# -amdgpu-kernarg-preload-count: gfx950 delivers the first 16 kernarg dwords in SGPRs at wave launch, so the
#   wave's first state loads do not sit behind a scalar-load round trip (Io is ordered hot-fields-first for it).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(f) > t for f in SOURCES + DEVICE_INCLUDES + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO_PATH
    cmd = [find_hipcc(), *HIPCC_FLAGS, "-I", os.path.join(ROOT, "include"), "-o", SO_PATH, *SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
