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


STAMP_PATH = SO_PATH + ".srchash"  # digest of what the library was built from (next to it, git-ignored like the .so)


def source_digest() -> str:
    """SHA-256 over the sources, headers, flags and this recipe: the library is current iff its stamp equals this
    (modification times say nothing -- a copied or checked-out tree can carry a newer-looking stale library)."""
    import hashlib

    h = hashlib.sha256()
    for f in SOURCES + DEVICE_INCLUDES + HEADERS:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def is_stale() -> bool:
    if not os.path.exists(SO_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as fh:
        return fh.read().strip() != source_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO_PATH
    cmd = [find_hipcc(), *HIPCC_FLAGS, "-I", os.path.join(ROOT, "include"), "-o", SO_PATH, *SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    digest = source_digest()
    subprocess.run(cmd, check=True)
    with open(STAMP_PATH, "w") as fh:
        fh.write(digest + "\n")
    return SO_PATH


# The checking build (mapf_kernels.inl: MAPF_CHK): same sources with -DMAPF_CHECK, reduced to the shapes the soak of the
# specialised kernels uses (groups of 4 and 8 lanes, 5 x 5 windows), so that it compiles in about a minute.  Selected
# per handle with MAPF_CHECK_BUILD=1 (_lib.load); tests/test_soak_gpu.py runs one soak on it.
CHECK_SO_PATH = os.path.join(CSRC, "libmapfstep_check.so")
CHECK_FLAGS = ["-DMAPF_CHECK", "-DMAPF_SMALL_SHAPES"]


def build_check(force: bool = False, verbose: bool = False) -> str:
    stamp = CHECK_SO_PATH + ".srchash"
    digest = source_digest() + "+check"
    if not force and os.path.exists(CHECK_SO_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return CHECK_SO_PATH
    cmd = [find_hipcc(), *HIPCC_FLAGS, *CHECK_FLAGS, "-I", os.path.join(ROOT, "include"), "-o", CHECK_SO_PATH, *SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    return CHECK_SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_check(force="--force" in sys.argv, verbose=True))
