#!/usr/bin/env python3
"""
Instruction counts of the compute-bound kernels, taken from the gfx950 ISA the library is built from (no GPU needed):
the float64 instructions a kernel issues per frame are what its FP64-VALU roofline is priced on ("report
HBM-write fraction and FP64-VALU fraction"; bench.py `roofline_fp64`).

    python tools/isa_count.py                 # the kernels of KERNELS -> JSON on stdout (profiles/r4_isa_counts.json)
    python tools/isa_count.py --loops pgx_scan.hip k_supersaw_wideILi4    # every loop of one kernel, to choose from

How.  `hipcc -S --cuda-device-only -DPGX_COUNT_STEADY` with build.py's flags gives the device assembly of a translation
unit with the kernels' rare branches compiled out (pgx_common.h: PGX_COLD / PGX_HOT -- a workgroup's first tile, the
guarded re-run after a singularity, the thread that captures the carried state).  LLVM annotates every basic block of
that assembly with the loop it belongs to ("in Loop: Header=BB31_35 Depth=2"); the instructions of a loop's blocks --
wherever the block placement put them -- are counted by class, inner loops added to the loops around them:
    f64        every v_*_f64 instruction and every conversion to or from float64 but the three below: full rate, one issue
               slot per lane (measured, tools/microbench/f64_rate.hip + experiments/README.md: v_fma / v_mul / v_add /
               v_rndne / v_floor and the float32 <-> float64 conversions all take 4.8 - 5.4 cycles per wave instruction)
    f64_slow   v_rcp_f64 / v_rsq_f64 / v_sqrt_f64: quarter rate (v_rcp_f64 measured at 17 cycles), four slots
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
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-Wno-unused-result", "-w",
         "-DPGX_COUNT_STEADY=1"]
F64_PEAK_SLOTS = 256 * 4 * 16 * 2.4e9

SLOW = re.compile(r"^v_(rcp|rsq|sqrt)_f64")
FULL = re.compile(r"^v_\w+_f64(_e32|_e64|_dpp|_sdwa)?$|^v_cvt_\w*f64|^v_cmpx?_\w+_f64")


def classify(op: str) -> str:
    if SLOW.match(op):
        return "f64_slow"
    if FULL.match(op) or (op.startswith("v_") and "_f64" in op):
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
    """{mangled name: [block]} with block = {"label", "loop" (innermost loop's header label or None), "header" (bool),
    "depth", "parents" [(label, depth)], "insts" [text]}."""
    out, cur, block = {}, None, None
    lines = asm.splitlines()
    i = 0
    while i < len(lines):
        raw = lines[i]
        s = raw.strip()
        i += 1
        if not s:
            continue
        m = re.match(r"^(_Z[\w$.]*):", s)
        if m:
            cur = out.setdefault(m.group(1), [])
            block = {"label": "entry", "loop": None, "header": False, "depth": 0, "parents": [], "insts": []}
            cur.append(block)
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(?:(\.LBB\w+):|; %bb\.(\d+):)\s*(;.*)?$", s)
        if m:
            label = m.group(1) or f"bb.{m.group(2)}"
            notes = [m.group(3) or ""]
            while i < len(lines) and re.match(r"^\s+;", lines[i]):       # the annotation's continuation lines
                notes.append(lines[i].strip())
                i += 1
            text = " ".join(notes)
            block = {"label": label, "loop": None, "header": False, "depth": 0, "parents": [], "insts": []}
            h = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text)
            if h:
                block.update(loop=label.replace(".L", ""), header=True, depth=int(h.group(1)))
                block["parents"] = [(a, int(b)) for a, b in re.findall(r"Parent Loop (BB\w+) Depth=(\d+)", text)]
            else:
                h = re.search(r"in Loop: Header=(BB\w+) Depth=(\d+)", text)
                if h:
                    block.update(loop=h.group(1), depth=int(h.group(2)))
            cur.append(block)
            continue
        if s.startswith((".", ";", "//")):
            continue
        block["insts"].append(s.split(";")[0].strip())
    return out


def loops(blocks):
    """{header: {"depth", "parent", "own" counts, "counts" (with inner loops)}} of one function."""
    found = {}
    for b in blocks:
        if b["loop"] is None:
            continue
        l = found.setdefault(b["loop"], {"head": b["loop"], "depth": b["depth"], "parent": None, "own": {}, "counts": {}})
        if b["header"]:
            l["depth"] = b["depth"]
            near = [p for p, d in b["parents"] if d == b["depth"] - 1]
            l["parent"] = near[-1] if near else None
        for text in b["insts"]:
            c = classify(text.split()[0])
            l["own"][c] = l["own"].get(c, 0) + 1
    for l in found.values():
        l["counts"] = dict(l["own"])
    for l in sorted(found.values(), key=lambda l: -l["depth"]):            # innermost first: add to every ancestor
        p = l["parent"]
        while p is not None and p in found:
            for c, v in l["own"].items():
                found[p]["counts"][c] = found[p]["counts"].get(c, 0) + v
            p = found[p]["parent"]
    return found


def slots(counts) -> float:
    return counts.get("f64", 0) + 4.0 * counts.get("f64_slow", 0)


# kernel -> (source, substring of the mangled name, how the steady-state body is chosen, units one lane makes per trip)
# "largest": the loop with the most float64 slots among those of the given nesting depth (0 = outermost).
# depth: LLVM's loop depth (1 = outermost) of the steady-state body; among the loops of that depth the one with the most
# float64 slots is taken.
KERNELS = {
    "k_biquad_settled<mono, staged, sine, 256>": dict(
        source="pgx_scan.hip", match="k_biquad_settledILb1ELb1ELb1ELi256", depth=2, units=16, unit="frame",
        why="the loop over a workgroup's tiles; a lane makes the 16 frames of its slot per tile (sine rotation, "
            "zero-state pass, DPP scan, carry-in pass)"),
    "k_biquad_settled<mono, staged, 512>": dict(
        source="pgx_scan.hip", match="k_biquad_settledILb1ELb1ELb0ELi512", depth=2, units=16, unit="frame",
        why="the loop over a workgroup's tiles; 16 frames per lane and tile"),
    "k_supersaw_wide<4>": dict(
        source="pgx_scan.hip", match="k_supersaw_wideILi4", depth=2, units=16, unit="oscillator-frame",
        why="tiles (outer) x voices (inner): one trip of the voice loop is one voice's 16 frames of a lane"),
    "k_blitsaw_biquad_wide<4, env>": dict(
        source="pgx_scan.hip", match="k_blitsaw_biquad_wideILi4ELb1", depth=1, units=16, unit="voice-frame",
        why="one voice per workgroup: the loop over its tiles, 16 frames per lane and tile (oscillator + filter)"),
    "k_voice_tiles<4, env>": dict(
        source="pgx_scan.hip", match="k_voice_tilesILi4ELb1", depth=1, units=16, unit="voice-frame",
        why="a workgroup's loop over its voices: one trip is one voice's 16 frames of a lane over the tile (anchor turn, "
            "oscillator, integrator scan, filter, x envelope, accumulate)"),
    "k_blitsaw_biquad_wide<4>": dict(
        source="pgx_scan.hip", match="k_blitsaw_biquad_wideILi4ELb0", depth=1, units=16, unit="voice-frame",
        why="one voice per workgroup: the loop over its tiles, 16 frames per lane and tile (oscillator + filter)"),
}


def kernel_report(name, spec, cache):
    if spec["source"] not in cache:
        cache[spec["source"]] = functions(assembly(spec["source"]))
    funcs = cache[spec["source"]]
    names = [n for n in funcs if spec["match"] in n]
    if not names:
        return {"error": f"no function matches {spec['match']}"}
    blocks = funcs[names[0]]
    total = {}
    for b in blocks:
        for text in b["insts"]:
            c = classify(text.split()[0])
            total[c] = total.get(c, 0) + 1
    ls = loops(blocks)
    cands = [l for l in ls.values() if l["depth"] == spec["depth"]] if spec["depth"] else []
    if cands:
        chosen = max(cands, key=lambda l: slots(l["counts"]))
    else:
        chosen = {"head": None, "counts": total, "depth": None}
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
                for l in loops(funcs[n]).values():
                    print("  " * l["depth"], f"{l['head']} depth {l['depth']} parent {l['parent']} slots "
                          f"{slots(l['counts']):.0f}", json.dumps(l["counts"]), "own", json.dumps(l["own"]))
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
