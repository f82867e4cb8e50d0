"""Build a pygmu2_amd PE graph (the HIP product path) from a golden-case SPEC
(see oracle/golden_cases.py for the format)."""

import pygmu2_amd as pg
from oracle.golden_cases import materialize_array

_SIMPLE = {
    "ConstantPE": pg.ConstantPE, "IdentityPE": pg.IdentityPE, "DiracPE": pg.DiracPE,
    "ArrayPE": pg.ArrayPE, "CropPE": pg.CropPE, "SinePE": pg.SinePE, "GainPE": pg.GainPE,
    "BlitSawPE": pg.BlitSawPE, "SuperSawPE": pg.SuperSawPE, "CombPE": pg.CombPE,
    "AdsrGatedPE": pg.AdsrGatedPE, "AdsrTriggeredPE": pg.AdsrTriggeredPE,
    "PeriodicGate": pg.PeriodicGate, "PeriodicTrigger": pg.PeriodicTrigger,
}


def build(spec):
    kind = spec["pe"]
    kw = {}
    for k, v in spec.items():
        if k == "pe":
            continue
        if isinstance(v, dict) and "pe" in v:
            kw[k] = build(v)
        elif isinstance(v, dict):
            kw[k] = materialize_array(v)
        elif k == "inputs":
            kw[k] = [build(s) for s in v]
        else:
            kw[k] = v
    if "extend_mode" in kw:
        kw["extend_mode"] = pg.ExtendMode(kw["extend_mode"])
    if kind in _SIMPLE:
        return _SIMPLE[kind](**kw)
    if kind == "MixPE":
        return pg.MixPE(*kw["inputs"])
    if kind == "BiquadPE":
        if "mode" in kw:
            kw["mode"] = pg.BiquadMode(kw["mode"])
        return pg.BiquadPE(**kw)
    if kind == "LadderPE":
        if "mode" in kw:
            kw["mode"] = pg.LadderMode(kw["mode"])
        return pg.LadderPE(**kw)
    if kind == "SVFilterPE":
        if "mode" in kw:
            kw["mode"] = pg.BiquadMode(kw["mode"])
        return pg.SVFilterPE(**kw)
    if kind == "EnvelopePE":
        if "mode" in kw:
            kw["mode"] = pg.DetectionMode(kw["mode"])
        return pg.EnvelopePE(**kw)
    if kind == "LoopPE":
        return pg.LoopPE(kw.pop("source"), **kw)
    if kind == "WindowPE":
        if "mode" in kw:
            kw["mode"] = pg.WindowMode(kw["mode"])
        return pg.WindowPE(**kw)
    if kind == "DynamicsPE":
        if "mode" in kw:
            kw["mode"] = pg.DynamicsMode(kw["mode"])
        return pg.DynamicsPE(**kw)
    if kind in ("CompressorPE", "LimiterPE", "ExpanderPE"):
        if "detection" in kw:
            kw["detection"] = pg.DetectionMode(kw["detection"])
        return getattr(pg, kind)(kw.pop("source"), **kw)
    if kind == "CachePE":
        return pg.CachePE(kw["source"])
    if kind == "TransformPE":
        return pg.TransformPE(kw["source"], func=pg.transforms.from_spec(kw["ops"]), name="ops")
    if kind == "SpatialPE":
        import os
        m = kw["method"]
        if m == "adapter":
            method = pg.SpatialAdapter(kw["channels"])
        elif m == "linear":
            method = pg.SpatialLinear(kw["azimuth"])
        elif m == "constant_power":
            method = pg.SpatialConstantPower(kw["azimuth"])
        else:
            method = pg.SpatialHRTF(kw["azimuth"], kw.get("elevation", 0.0),
                                    kemar_dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                           "kemar"))
        return pg.SpatialPE(kw["source"], method=method)
    if kind == "DelayPE":
        if "interpolation" in kw:
            kw["interpolation"] = pg.InterpolationMode(kw["interpolation"])
        return pg.DelayPE(**kw)
    if kind == "PiecewisePE":
        kw["points"] = [(int(t), float(v)) for t, v in kw["points"]]
        return pg.PiecewisePE(**kw)
    if kind == "TriggerRestartPE":
        return pg.TriggerRestartPE(kw["trigger"], kw["src"])
    if kind == "ReverbPE":
        return pg.ReverbPE(kw.pop("source"), kw.pop("ir"), kw.pop("mix", 0.5), **kw)
    if kind == "ConvolvePE":
        return pg.ConvolvePE(kw.pop("src"), kw.pop("fir"), **kw)
    raise KeyError(kind)


def run_case(case):
    """Render every block of a case through a started NullRenderer graph; list of arrays."""
    pg.set_sample_rate(case["sr"])
    pe = build(case["graph"])
    r = pg.NullRenderer(sample_rate=case["sr"])
    r.set_source(pe)
    r.start()
    outs = [pe.render(int(s), int(n)).data for s, n in case["blocks"]]
    r.stop()
    return outs
