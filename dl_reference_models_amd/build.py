"""Builds libmapfstep.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m dl_reference_models_amd.build [--force] [--variant NAME ...]

The library is one host unit (csrc/mapf_step.hip: the C ABI) plus LAUNCH units (csrc/mapf_launch.hip compiled once per
kernel group: a prebuilt specialisation, the runtime-config kernels of one group width x window-mask width, the
single-agent kernels of one group width; csrc/mapf_engine.h).  The units compile in parallel, one hipcc per host core, and
are linked into one shared object: a cold build of the shipped library AND the checking build takes about as long as
the slowest unit instead of the sum of all of them (round 3: two serial compiles of one translation unit, 6 min 20 s).

hipcc cross-compiles without a GPU; the built .so travels with the source tree (git-ignored).  Objects are cached per
unit under csrc/_obj/<variant>/ and keyed by a digest of sources + flags, so only what changed is recompiled.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libmapfstep.so")
HOST_SOURCE = os.path.join(CSRC, "mapf_step.hip")
LAUNCH_SOURCE = os.path.join(CSRC, "mapf_launch.hip")
SOURCES = [HOST_SOURCE, LAUNCH_SOURCE]
DEVICE_INCLUDES = [os.path.join(CSRC, "mapf_kernels.inl"), os.path.join(CSRC, "mapf_engine.h")]
HEADERS = [os.path.join(ROOT, "include", "mapf_step.h")]

# NOTE: no -ffast-math -- goal_delta needs the correctly rounded fp32 divide.
# -amdgpu-kernarg-preload-count: gfx950 delivers the first 16 kernarg dwords in SGPRs at wave launch, so the
#   wave's first state loads do not sit behind a scalar-load round trip (Io is ordered hot-fields-first for it).
COMPILE_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
LINK_FLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
HIPCC_FLAGS = COMPILE_FLAGS + LINK_FLAGS  # (kept: part of the digest; tools print it)

ALL_LPE = (4, 8, 16, 32, 64)
ALL_MW = (32, 64, 128)


def _units(specials, runtime, cte):
    u = [("host", HOST_SOURCE, [])]
    u += [(f"special_{k}", LAUNCH_SOURCE, [f"-DMAPF_TU_SPECIAL={k}"]) for k in specials]
    u += [(f"runtime_{l}_{mw}", LAUNCH_SOURCE, [f"-DMAPF_TU_LPE={l}", f"-DMAPF_TU_MW={mw}"]) for l, mw in runtime]
    u += [(f"cte_{l}", LAUNCH_SOURCE, [f"-DMAPF_TU_CTE={l}"]) for l in cte]
    return u


# name -> (output path, extra flags for every unit, units).  The reduced variants hold what mapf_engine.h's MAPF_FOR_LPE /
# MAPF_FOR_MW / MAPF_SPECIALIZATIONS leave under the same define (mapf_create refuses everything else).
VARIANTS = {
    "full": (SO_PATH, [], _units(range(1, 7), [(l, mw) for l in ALL_LPE for mw in ALL_MW], ALL_LPE)),
    # The checking build (mapf_kernels.inl: MAPF_CHK): -DMAPF_CHECK, reduced to the shapes the soak of the specialised
    # kernels uses (groups of 4 and 8 lanes, 5 x 5 windows).  Selected per handle with MAPF_CHECK_BUILD=1 (_lib.load);
    # tests/test_soak_gpu.py runs soaks on it.
    "check": (os.path.join(CSRC, "libmapfstep_check.so"), ["-DMAPF_CHECK", "-DMAPF_SMALL_SHAPES"],
              _units((1, 2, 4, 5, 6), [(4, 32), (8, 32), (16, 32), (4, 64), (8, 64), (16, 64)], ())),
    # development builds (into build_diag/, never shipped): one shape each, -DMAPF_DEV turns the environment knobs on
    "dev_c3": (os.path.join(ROOT, "build_diag", "libdev.so"), ["-DMAPF_DEV", "-DMAPF_DEV_C3"], _units((1,), [(8, 32)], ())),
    "dev_c5": (os.path.join(ROOT, "build_diag", "libc5.so"), ["-DMAPF_DEV", "-DMAPF_DEV_C5"], _units((3,), [(64, 32)], ())),
    "dev_n16": (os.path.join(ROOT, "build_diag", "libn16.so"), ["-DMAPF_DEV", "-DMAPF_DEV_N16"], _units((6,), [(16, 64)], ())),
    "dev_cte": (os.path.join(ROOT, "build_diag", "libcte.so"), ["-DMAPF_DEV", "-DMAPF_DEV_CTE"],
                _units((1,), [(8, 32), (64, 32)], (8, 64))),
}
CHECK_SO_PATH = VARIANTS["check"][0]
CHECK_FLAGS = VARIANTS["check"][1]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _files_digest() -> "hashlib._Hash":
    h = hashlib.sha256()
    for f in SOURCES + DEVICE_INCLUDES + HEADERS:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h


def source_digest(variant: str = "full", extra=()) -> str:
    """SHA-256 over the sources, headers, flags and the unit list: the library is current iff its stamp equals this
    (modification times say nothing -- a copied or checked-out tree can carry a newer-looking stale library)."""
    _so, flags, units = VARIANTS[variant]
    h = _files_digest()
    h.update(" ".join(HIPCC_FLAGS + list(flags) + list(extra)).encode())
    h.update(repr([(n, os.path.basename(s), d) for n, s, d in units]).encode())
    return h.hexdigest()


STAMP_PATH = SO_PATH + ".srchash"  # digest of what the library was built from (next to it, git-ignored like the .so)


def is_stale(variant: str = "full", so_path: str | None = None, extra=()) -> bool:
    so = so_path or VARIANTS[variant][0]
    stamp = so + ".srchash"
    if not os.path.exists(so) or not os.path.exists(stamp):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != source_digest(variant, extra)


def _unit_jobs(variant, extra, obj_dir):
    """[(object path, digest, command)] of a variant's units; every unit includes every header, so a unit's digest is the
    files' digest + its own command line."""
    _so, flags, units = VARIANTS[variant]
    base = _files_digest().hexdigest()
    hipcc = find_hipcc()
    jobs = []
    for name, src, defs in units:
        obj = os.path.join(obj_dir, name + ".o")
        cmd = [hipcc, *COMPILE_FLAGS, *flags, *extra, *defs, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        digest = hashlib.sha256((base + " ".join(cmd[1:])).encode()).hexdigest()
        jobs.append((obj, digest, cmd))
    return jobs


def _compile(job, verbose):
    obj, digest, cmd = job
    stamp = obj + ".srchash"
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return obj, 0.0
    t0 = time.perf_counter()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    return obj, time.perf_counter() - t0


def default_jobs() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def build_variants(names, force: bool = False, verbose: bool = False, extra=(), out=None, jobs: int | None = None):
    """Builds the named variants with ALL their units in one pool (the shipped library and the checking build share the
    cores instead of queueing behind each other).  extra: flags for every unit (e.g. -DMAPF_STAMPS); out: output path
    override when a single variant is built.  Returns the paths."""
    names = list(names)
    assert out is None or len(names) == 1
    todo, outs = [], []
    for v in names:
        so = out or VARIANTS[v][0]
        outs.append(so)
        if force or is_stale(v, so, extra):
            obj_dir = os.path.join(CSRC, "_obj", v + ("-" + hashlib.sha256(" ".join(extra).encode()).hexdigest()[:8] if extra else ""))
            os.makedirs(obj_dir, exist_ok=True)
            os.makedirs(os.path.dirname(so), exist_ok=True)
            todo.append((v, so, _unit_jobs(v, list(extra), obj_dir)))
    if not todo:
        return outs
    flat = [j for _v, _so, js in todo for j in js]
    # slowest units first (the runtime-config kernels of the wide groups, the specialisations), so the pool drains evenly
    flat.sort(key=lambda j: (0 if "runtime_64" in j[0] or "runtime_32" in j[0] else 1 if "special" in j[0] else 2))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(jobs or default_jobs()) as ex:
        times = dict(ex.map(lambda j: _compile(j, verbose), flat))
    for v, so, js in todo:
        digest = source_digest(v, extra)
        cmd = [find_hipcc(), *LINK_FLAGS, "-o", so, *[o for o, _d, _c in js]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(so + ".srchash", "w") as fh:
            fh.write(digest + "\n")
    if verbose:
        slow = sorted(times.items(), key=lambda kv: -kv[1])[:5]
        print(f"[build] {len(flat)} units in {time.perf_counter() - t0:.0f} s wall; slowest: "
              + ", ".join(f"{os.path.basename(o)} {t:.0f} s" for o, t in slow), flush=True)
    return outs


def build(force: bool = False, verbose: bool = False) -> str:
    return build_variants(["full"], force, verbose)[0]


def build_check(force: bool = False, verbose: bool = False) -> str:
    return build_variants(["check"], force, verbose)[0]


def build_all(force: bool = False, verbose: bool = False):
    """The shipped library and the checking build, their units compiled side by side."""
    return build_variants(["full", "check"], force, verbose)


if __name__ == "__main__":
    argv = sys.argv[1:]
    names = [argv[i + 1] for i, a in enumerate(argv) if a == "--variant"] or ["full", "check"]
    extra = [a for a in argv if a.startswith("-D")]
    out = next((argv[i + 1] for i, a in enumerate(argv) if a == "--out"), None)
    for path in build_variants(names, force="--force" in argv, verbose=True, extra=extra, out=out):
        print(path)
