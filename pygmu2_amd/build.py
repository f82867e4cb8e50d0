"""
Build libpygmu_hip.so (the HIP render library) in-tree with hipcc for gfx950.

    python -m pygmu2_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU, so this also runs in the
CPU-only build container.  The library is built next to this file so that it travels
with the source tree to the GPU box.
"""

from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("PGX_LIB_PATH") or os.path.join(PKG_DIR, "libpygmu_hip.so")

SOURCES = [
    "pgx_runtime.hip",
    "pgx_elementwise.hip",
    "pgx_scan.hip",
    "pgx_seq.hip",
    "pgx_comb.hip",
    "pgx_adsr.hip",
    "pgx_convolve.hip",
    "pgx_lookup.hip",
    "pgx_dynamics.hip",
    "pgx_fftconv.hip",
    "pgx_comm.hip",
]

# -ffp-contract=off: the parity contract is "same float64 operation order as the reference's
# numpy/scipy code"; fused multiply-adds would change roundings.  Kernels that want an FMA ask
# for one explicitly.
FLAGS = [
    "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
    "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps_common():
    return [os.path.join(CSRC, "pgx_common.h"), os.path.join(ROOT, "include", "pygmu_hip.h"),
            os.path.abspath(__file__)]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB_PATH, [os.path.join(CSRC, s) for s in SOURCES] + _deps_common())


FAST_SRC = os.path.join(CSRC, "_fast.c")
FAST_LIB = os.path.join(PKG_DIR, "_fast.so")


def build_fast(force: bool = False, verbose: bool = True) -> str | None:
    """pygmu2_amd/_fast.so: the hot host paths (render()'s window exits, Snippet.__del__) as a CPython extension, built
    with the C compiler against this interpreter's headers.  Host glue: the package runs without it."""
    import sysconfig
    if not force and not _stale(FAST_LIB, [FAST_SRC, os.path.abspath(__file__)]):
        return FAST_LIB
    include = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(include, "Python.h")):
        if verbose:
            print("[pygmu2_amd.build] no Python.h: _fast.so not built (the Python paths are used)", flush=True)
        return None
    cc = os.environ.get("CC") or "gcc"
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I" + include, FAST_SRC, "-o", FAST_LIB]
    if verbose:
        print("[pygmu2_amd.build]", " ".join(cmd[-3:]), flush=True)
    subprocess.check_call(cmd)
    return FAST_LIB


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source (one object per translation unit, in parallel, only the stale ones) and
    link pygmu2_amd/libpygmu_hip.so; returns its path."""
    build_fast(force, verbose)
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(CSRC, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    extra = os.environ.get("PGX_EXTRA_FLAGS", "").split()       # experiments: -DPGX_... switches
    tag = os.path.join(obj_dir, "flags.txt")
    flag_text = " ".join(FLAGS + extra)
    if not os.path.exists(tag) or open(tag).read() != flag_text:
        force = True
    compile_flags = [f for f in FLAGS if f != "-shared"] + extra
    jobs, objs = [], []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(obj_dir, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + _deps_common()):
            jobs.append([_hipcc()] + compile_flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                                                      "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[pygmu2_amd.build]", " ".join(cmd[-4:]), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print("[pygmu2_amd.build] link", LIB_PATH, flush=True)
    subprocess.check_call(link)
    with open(tag, "w") as f:
        f.write(flag_text)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
