#!/usr/bin/env python3
"""
Instruction counts of the compute-bound kernels, taken from the gfx950 ISA the library is built from (no GPU needed):
the float64 instructions a kernel issues per frame are what its FP64-VALU roofline is priced on ("report
HBM-write fraction and FP64-VALU fraction"; bench.py `roofline_fp64`).

    python tools/isa_count.py                 # the kernels of KERNELS -> JSON on stdout (profiles/r4_isa_counts.json)
    python tools/isa_count.py --loops pgx_scan.hip k_supersaw_wideILi4    # every loop of one kernel, to choose from

How.  `hipcc -S --cuda-device-only` with build.py's flags gives the device assembly of a translation unit.  Inside a
kernel a loop is a branch to an earlier label; loops nest by containment.  For every loop the instructions between its
head label and its back edge are counted by class:
    f64        v_fma / v_mul / v_add / v_min / v_max / v_cmp*_f64: full rate, one issue slot per lane
    f64_slow   v_rcp / v_rsq / v_sqrt / v_rndne / v_floor / v_ceil / v_trunc / v_fract / v_frexp* / v_ldexp /
               v_div_* / v_trig_preop / v_cvt to or from f64: quarter rate (tools/microbench/f64_rate.hip), four slots
    valu32     every other v_* instruction (moves, selects, integer, DPP, float32): half a float64 slot each on the
               SIMD-32 (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles, a float64 one over 4)
    salu / lds / vmem / other
The peak is 256 CUs x 4 SIMDs x 16 float64 lanes x 2.4 GHz = 39.3 T lane-slots/s (78.6 TFLOP/s counting an FMA as two).
A kernel's `slots_per_unit` is (f64 + 4 f64_slow) of its steady-state loop body divided by the units (frames,
oscillator-frames, samples) one lane produces per trip -- the body and the units per trip are named in KERNELS with the
reason; `slots_per_unit_all_valu` adds valu32 / 2.  Code outside that loop (prologues, first-tile paths, rare branches)
is not counted: the figure is the kernel's steady state, a lower bound on what it issues.
"""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pygmu2_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result", "-w"]
F64_PEAK_SLOTS = 256 * 4 * 16 * 2.4e9

SLOW = re.compile(r"^v_(rcp|rsq|sqrt|rndne|floor|ceil|trunc|fract|frexp_mant|frexp_exp_i32|ldexp|div_scale|div_fmas|"
                  r"div_fixup|trig_preop)_f64|^v_cvt_.*f64")
FULL = re.compile(r"^v_(fma|mul|add|min|max|max_num|min_num|fmac)_f64|^v_cmp[a-z_]*_f64|^v_cmpx[a-z_]*_f64")


def classify(op: str) -> str:
    if SLOW.match(op):
        return "f64_slow"
    if FULL.match(op) or (op.startswith("v_") and op.endswith("_f64")):
        return "f64"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu32"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def assembly(source: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                                                 "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, source)]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(out) as f:
            return f.read()


def functions(asm: str):
    """{mangled name: [(kind, text)]} with kind 'label' or 'inst'."""
    out, cur = {}, None
    for line in asm.splitlines():
        s = line.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^(_Z[\w$.]*):", s)
        if m and not s.startswith(".L"):
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            cur.append(("label", m.group(1)))
            continue
        if s.startswith("."):
            continue
        cur.append(("inst", s.split(";")[0].strip()))
    return out


def loops(body):
    """Loops of one function: [{head, start, end, counts, depth, own}] sorted by start; indices are positions in `body`."""
    pos = {text: i for i, (kind, text) in enumerate(body) if kind == "label"}
    found = {}
    for i, (kind, text) in enumerate(body):
        if kind != "inst":
            continue
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", text)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            head = m.group(1)
            found[head] = max(found.get(head, -1), i)                  # the outermost back edge to this head
    result = []
    for head, end in found.items():
        start = pos[head]
        counts = {}
        for kind, text in body[start:end + 1]:
            if kind == "inst":
                c = classify(text.split()[0])
                counts[c] = counts.get(c, 0) + 1
        result.append({"head": head, "start": start, "end": end, "counts": counts})
    result.sort(key=lambda l: (l["start"], -l["end"]))
    for l in result:
        inside = [m for m in result if m is not l and m["start"] >= l["start"] and m["end"] <= l["end"]]
        l["depth"] = sum(1 for m in result if m is not l and m["start"] <= l["start"] and m["end"] >= l["end"])
        own = dict(l["counts"])
        for m in inside:
            if not any(k is not m and k is not l and k["start"] <= m["start"] and k["end"] >= m["end"] for k in inside):
                for c, v in m["counts"].items():                         # direct children only
                    own[c] = own.get(c, 0) - v
        l["own"] = own
    return result


def slots(counts) -> float:
    return counts.get("f64", 0) + 4.0 * counts.get("f64_slow", 0)


# kernel -> (source, substring of the mangled name, how the steady-state body is chosen, units one lane makes per trip)
# "largest": the loop with the most float64 slots among those of the given nesting depth (0 = outermost).
KERNELS = {
    "k_biquad_settled<mono, staged, sine, 256>": dict(
        source="pgx_scan.hip", match="k_biquad_settledILb1ELb1ELb1ELi256", depth=0, units=16, unit="frame",
        why="the loop over a workgroup's tiles; a lane makes the 16 frames of its slot per tile (sine rotation, "
            "zero-state pass, DPP scan, carry-in pass)"),
    "k_biquad_settled<mono, staged, 512>": dict(
        source="pgx_scan.hip", match="k_biquad_settledILb1ELb1ELb0ELi512", depth=0, units=16, unit="frame",
        why="the loop over a workgroup's tiles; 16 frames per lane and tile"),
    "k_supersaw_wide<4>": dict(
        source="pgx_scan.hip", match="k_supersaw_wideILi4", depth=1, units=16, unit="oscillator-frame",
        why="tiles (outer) x voices (inner): one trip of the voice loop is one voice's 16 frames of a lane"),
    "k_blitsaw_biquad_wide<4, env>": dict(
        source="pgx_scan.hip", match="k_blitsaw_biquad_wideILi4ELb1", depth=0, units=16, unit="voice-frame",
        why="one voice per workgroup: the loop over its tiles, 16 frames per lane and tile (oscillator + filter)"),
    "k_blitsaw_biquad_wide<4>": dict(
        source="pgx_scan.hip", match="k_blitsaw_biquad_wideILi4ELb0", depth=0, units=16, unit="voice-frame",
        why="one voice per workgroup: the loop over its tiles, 16 frames per lane and tile (oscillator + filter)"),
    "k_sine": dict(
        source="pgx_elementwise.hip", match="k_sine", depth=None, units=4, unit="frame",
        why="grid-stride body: four frames per lane and trip (no loop: the whole kernel is the body)"),
}


def kernel_report(name, spec, cache):
    if spec["source"] not in cache:
        cache[spec["source"]] = functions(assembly(spec["source"]))
    funcs = cache[spec["source"]]
    names = [n for n in funcs if spec["match"] in n]
    if not names:
        return {"error": f"no function matches {spec['match']}"}
    body = funcs[names[0]]
    total = {}
    for kind, text in body:
        if kind == "inst":
            c = classify(text.split()[0])
            total[c] = total.get(c, 0) + 1
    ls = loops(body)
    if spec["depth"] is None or not ls:
        chosen = {"head": None, "counts": total, "depth": None}
    else:
        cands = [l for l in ls if l["depth"] == spec["depth"]] or ls
        chosen = max(cands, key=lambda l: slots(l["counts"]))
    c = chosen["counts"]
    units = spec["units"]
    return {"function": names[0], "loop_head": chosen["head"], "loop_depth": chosen["depth"], "unit": spec["unit"],
            "units_per_lane_per_trip": units, "why": spec["why"], "body_counts": c, "whole_kernel_counts": total,
            "f64_slots_per_trip": slots(c),
            "slots_per_unit": round(slots(c) / units, 3),
            "slots_per_unit_all_valu": round((slots(c) + 0.5 * c.get("valu32", 0)) / units, 3),
            "f64_instructions_per_unit": round((c.get("f64", 0) + c.get("f64_slow", 0)) / units, 3)}


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--loops":
        funcs = functions(assembly(sys.argv[2]))
        for n in funcs:
            if sys.argv[3] in n:
                print(n)
                for l in loops(funcs[n]):
                    print("  " * l["depth"], f"{l['head']} [{l['start']}..{l['end']}] slots {slots(l['counts']):.0f}",
                          json.dumps(l["counts"]), "own", json.dumps(l["own"]))
        return
    cache = {}
    out = {"peak_f64_lane_slots_per_s": F64_PEAK_SLOTS,
           "weights": {"f64": 1, "f64_slow": 4, "valu32": 0.5},
           "flags": " ".join(FLAGS), "kernels": {}}
    for name, spec in KERNELS.items():
        out["kernels"][name] = kernel_report(name, spec, cache)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
