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
    "pgx_adsr.hip",
    "pgx_convolve.hip",
    "pgx_lookup.hip",
    "pgx_dynamics.hip",
    "pgx_fftconv.hip",
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


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps += [os.path.join(CSRC, "pgx_common.h"), os.path.join(ROOT, "include", "pygmu_hip.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into pygmu2_amd/libpygmu_hip.so; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    extra = os.environ.get("PGX_EXTRA_FLAGS", "").split()       # experiments: -DPGX_... switches
    cmd = [_hipcc()] + FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                                "-o", LIB_PATH] + srcs
    if verbose:
        print("[pygmu2_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
