#!/usr/bin/env python3
"""
oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY; runs ONLY in the build container,
where the read-only reference tree exists at /root/reference.

Imports the reference's PE modules (bare-package loader: the package __init__ pulls
soundfile/sounddevice which are absent; numba is replaced by a pass-through decorator
so the *_numba kernel bodies run as plain Python, i.e. the block-size-invariant numba
semantics), builds every case of oracle/golden_cases.py with
the reference's own classes, renders the listed blocks through a started
NullRenderer graph and writes

    tests/golden/cases.json      (case list: graph SPECs, blocks, sample rate)
    tests/golden/golden.npz      (float32 outputs, key "<case>/<block index>")

Only data (inputs are regenerated from the SPEC, outputs are stored) goes into the
repository; no reference source text is copied.

Usage:  PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py
"""

from __future__ import annotations

import importlib
import json
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/src/pygmu2"

from oracle.golden_cases import cases, materialize_array  # noqa: E402


def load_reference():
    logging.disable(logging.CRITICAL)

    def _passthrough(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb = types.ModuleType("numba")
    nb.jit = _passthrough
    nb.njit = _passthrough
    sys.modules["numba"] = nb
    # spatial_pe imports soundfile (absent) at module level; only SpatialHRTF._load_ir calls it, and the
    # HRTF cases pre-fill the instance's IR cache from the fixture WAVs, so an empty module is enough
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    pkg = types.ModuleType("pygmu2")
    pkg.__path__ = [REF]
    sys.modules["pygmu2"] = pkg
    mods = {}
    for name in ("config", "extent", "snippet", "null_renderer", "constant_pe", "identity_pe",
                 "dirac_pe", "array_pe", "crop_pe", "sine_pe", "gain_pe", "mix_pe", "biquad_pe",
                 "blit_saw_pe", "super_saw_pe", "ladder_pe", "comb_pe", "adsr_pe",
                 "periodic_gate", "periodic_trigger", "convolve_pe", "svfilter_pe", "envelope_pe",
                 "transform_pe", "wavetable_pe", "delay_pe", "piecewise_pe", "trigger_restart_pe", "cache_pe",
                 "reverb_pe", "assets", "spatial_pe", "loop_pe", "window_pe", "conversions", "dynamics_pe",
                 "compressor_pe"):
        mods[name] = importlib.import_module(f"pygmu2.{name}")
    return mods


def build(spec, M):
    """SPEC -> reference PE instance."""
    kind = spec["pe"]
    kw = {}
    for k, v in spec.items():
        if k == "pe":
            continue
        if isinstance(v, dict) and "pe" in v:
            kw[k] = build(v, M)
        elif isinstance(v, dict):
            kw[k] = materialize_array(v)
        elif k == "inputs":
            kw[k] = [build(s, M) for s in v]
        else:
            kw[k] = v
    E = M["extent"].ExtendMode
    if "extend_mode" in kw:
        kw["extend_mode"] = E(kw["extend_mode"])
    if kind == "ConstantPE":
        return M["constant_pe"].ConstantPE(**kw)
    if kind == "IdentityPE":
        return M["identity_pe"].IdentityPE(**kw)
    if kind == "DiracPE":
        return M["dirac_pe"].DiracPE(**kw)
    if kind == "ArrayPE":
        return M["array_pe"].ArrayPE(**kw)
    if kind == "CropPE":
        return M["crop_pe"].CropPE(**kw)
    if kind == "SinePE":
        return M["sine_pe"].SinePE(**kw)
    if kind == "GainPE":
        return M["gain_pe"].GainPE(**kw)
    if kind == "MixPE":
        return M["mix_pe"].MixPE(*kw["inputs"])
    if kind == "BiquadPE":
        if "mode" in kw:
            kw["mode"] = M["biquad_pe"].BiquadMode(kw["mode"])
        return M["biquad_pe"].BiquadPE(**kw)
    if kind == "BlitSawPE":
        return M["blit_saw_pe"].BlitSawPE(**kw)
    if kind == "SuperSawPE":
        return M["super_saw_pe"].SuperSawPE(**kw)
    if kind == "LadderPE":
        if "mode" in kw:
            kw["mode"] = M["ladder_pe"].LadderMode(kw["mode"])
        return M["ladder_pe"].LadderPE(**kw)
    if kind == "CombPE":
        return M["comb_pe"].CombPE(**kw)
    if kind == "AdsrGatedPE":
        return M["adsr_pe"].AdsrGatedPE(**kw)
    if kind == "AdsrTriggeredPE":
        return M["adsr_pe"].AdsrTriggeredPE(**kw)
    if kind == "PeriodicGate":
        return M["periodic_gate"].PeriodicGate(**kw)
    if kind == "PeriodicTrigger":
        return M["periodic_trigger"].PeriodicTrigger(**kw)
    if kind == "ConvolvePE":
        return M["convolve_pe"].ConvolvePE(kw.pop("src"), kw.pop("fir"), **kw)
    if kind == "SVFilterPE":
        if "mode" in kw:
            kw["mode"] = M["biquad_pe"].BiquadMode(kw["mode"])
        return M["svfilter_pe"].SVFilterPE(**kw)
    if kind == "EnvelopePE":
        if "mode" in kw:
            kw["mode"] = M["envelope_pe"].DetectionMode(kw["mode"])
        return M["envelope_pe"].EnvelopePE(**kw)
    if kind == "SpatialPE":
        sp = M["spatial_pe"]
        method = kw["method"]
        if method == "adapter":
            meth = sp.SpatialAdapter(kw["channels"])
        elif method == "linear":
            meth = sp.SpatialLinear(kw["azimuth"])
        elif method == "constant_power":
            meth = sp.SpatialConstantPower(kw["azimuth"])
        else:
            meth = sp.SpatialHRTF(kw["azimuth"], kw.get("elevation", 0.0))
            name = sp.SpatialHRTF.hrtf_filename_for(meth.azimuth, meth.elevation)
            import wave
            with wave.open(os.path.join(ROOT, "tests", "golden", "kemar", name), "rb") as w:
                pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
                # libsndfile's PCM16 -> float32 (x / 32768), what sf.read(dtype="float32") returns
                meth._ir_cache[name] = ((pcm.astype(np.float32) * np.float32(1.0 / 32768.0)), w.getframerate())
        return sp.SpatialPE(kw["source"], method=meth)
    if kind == "DelayPE":
        if "interpolation" in kw:
            kw["interpolation"] = M["wavetable_pe"].InterpolationMode(kw["interpolation"])
        return M["delay_pe"].DelayPE(**kw)
    if kind == "PiecewisePE":
        kw["points"] = [(int(t), float(v)) for t, v in kw["points"]]
        return M["piecewise_pe"].PiecewisePE(**kw)
    if kind == "TriggerRestartPE":
        return M["trigger_restart_pe"].TriggerRestartPE(kw["trigger"], kw["src"])
    if kind == "ReverbPE":
        return M["reverb_pe"].ReverbPE(kw.pop("source"), kw.pop("ir"), kw.pop("mix", 0.5), **kw)
    if kind == "LoopPE":
        return M["loop_pe"].LoopPE(kw.pop("source"), **kw)
    if kind == "WindowPE":
        if "mode" in kw:
            kw["mode"] = M["window_pe"].WindowMode(kw["mode"])
        return M["window_pe"].WindowPE(**kw)
    if kind == "DynamicsPE":
        if "mode" in kw:
            kw["mode"] = M["dynamics_pe"].DynamicsMode(kw["mode"])
        return M["dynamics_pe"].DynamicsPE(**kw)
    if kind in ("CompressorPE", "LimiterPE", "ExpanderPE"):
        if "detection" in kw:
            kw["detection"] = M["envelope_pe"].DetectionMode(kw["detection"])
        return getattr(M["compressor_pe"], kind)(kw.pop("source"), **kw)
    if kind == "CachePE":
        return M["cache_pe"].CachePE(kw["source"])
    if kind == "TransformPE":
        return M["transform_pe"].TransformPE(kw["source"], func=numpy_func(kw["ops"]), name="ops")
    raise KeyError(kind)


def numpy_func(ops):
    """The plain numpy callable a reference user would pass to TransformPE for this op list."""
    def f(v):
        for op in ops:
            name = op[0]
            if name == "affine":
                v = op[2] + op[1] * v
            elif name == "clip":
                v = np.clip(v, op[1], op[2])
            elif name == "sqrt":
                v = v ** 0.5
            elif name == "square":
                v = v ** 2
            elif name == "abs":
                v = np.abs(v)
            elif name == "tanh":
                v = np.tanh(v)
            elif name == "one_minus":
                v = 1.0 - v
            else:
                raise KeyError(name)
        return v
    return f


def has_kind(spec, kind):
    if isinstance(spec, dict):
        if spec.get("pe") == kind:
            return True
        return any(has_kind(v, kind) for v in spec.values())
    if isinstance(spec, list):
        return any(has_kind(v, kind) for v in spec)
    return False


def main():
    M = load_reference()
    out = {}
    case_list = cases()
    for case in case_list:
        M["config"].set_sample_rate(case["sr"])
        pe = build(case["graph"], M)
        r = M["null_renderer"].NullRenderer(sample_rate=case["sr"])
        r.set_source(pe)
        # A reference ConvolvePE cannot be start()ed (its _reset_state drops the tail
        # that _ensure_filter_prepared never re-creates); the
        # reference's own tests render it un-started, so do the same.
        if not (has_kind(case["graph"], "ConvolvePE") or has_kind(case["graph"], "ReverbPE")):
            r.start()
        for i, (s, n) in enumerate(case["blocks"]):
            data = pe.render(int(s), int(n)).data
            assert data.dtype == np.float32 and data.shape[0] == n, (case["name"], data.dtype, data.shape)
            if i in case["keep"]:
                out[f"{case['name']}/{i}"] = np.ascontiguousarray(data)
        print(f"{case['name']:36s} blocks={len(case['blocks'])}")
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    np.savez_compressed(os.path.join(gdir, "golden.npz"), **out)
    with open(os.path.join(gdir, "cases.json"), "w") as f:
        json.dump(case_list, f, indent=0)
    sz = os.path.getsize(os.path.join(gdir, "golden.npz"))
    print(f"{len(case_list)} cases, {len(out)} blocks, golden.npz = {sz / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
