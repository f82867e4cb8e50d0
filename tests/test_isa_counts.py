"""CPU: the float64 instruction counts bench.py prices the compute-bound kernels with (profiles/r4_isa_counts.json) are
the ones tools/isa_count.py takes from the gfx950 ISA of the current sources (hipcc cross-compiles: no GPU needed)."""

import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_committed_isa_counts_match_the_sources():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_count.py")], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    fresh = json.loads(p.stdout)
    with open(os.path.join(ROOT, "profiles", "r4_isa_counts.json")) as f:
        kept = json.load(f)
    assert set(fresh["kernels"]) == set(kept["kernels"])
    for name, k in fresh["kernels"].items():
        assert "error" not in k, (name, k)
        assert k["body_counts"] == kept["kernels"][name]["body_counts"], name
        assert k["slots_per_unit"] == kept["kernels"][name]["slots_per_unit"], name
        # a steady-state body issues float64 work and little else in the way of cold code
        assert k["body_counts"].get("f64", 0) > 100 and k["slots_per_unit"] > 5, name


def test_fp64_roofline_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    r = bench.fp64_roofline("k_supersaw_wide<4>", 512 * 7 * 48000, 0.150)
    k = bench.isa_counts()["kernels"]["k_supersaw_wide<4>"]
    want = 512 * 7 * 48000 / 0.150e-3 * k["slots_per_unit"] / 1e12
    assert abs(r["achieved"] - want) < 1e-3 * want and abs(r["frac"] - want / 39.3216) < 1e-3
    assert r["bound"] == "fp64_valu" and r["peak"] == 39.32
    assert bench.fp64_roofline("no such kernel", 1, 1.0) is None
