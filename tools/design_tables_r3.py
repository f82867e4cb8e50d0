#!/usr/bin/env python3
"""Markdown tables for DESIGN.md section 7 from a committed bench line and the previous round's beside it:
python tools/design_tables_r3.py [profiles/r4_bench.json [profiles/r3_bench.json]] > /tmp/tables.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
new = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r3_bench.json")
old = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r2_bench.json" if "r3" in new else "r3_bench.json")
d = json.load(open(new))
o = json.load(open(old))


def n(v, digits=0):
    return "" if v is None else f"{v:,.{digits}f}".replace(",", " ")


print("### benchmark_pes.py suite (44 100-frame renders, 5 + 50): Msamples/s\n")
print("| config | sync | pipelined | block by block | CPU oracle | pipelined / CPU | previous round, pipelined |")
print("|---|---|---|---|---|---|---|")
for name, row in d["suite"]["rows"].items():
    old = o["suite"]["rows"].get(name, {})
    cpu = row.get("cpu")
    print(f"| {name} | {n(row['sync'])} | {n(row['pipelined'])} | {n(row['block_by_block'])} | {n(cpu, 1) if cpu else ''} | "
          f"{n(row['pipelined'] / cpu) + 'x' if cpu else ''} | {n(old.get('pipelined'))} |")
print("\n### north_star PEs outside that suite (same protocol)\n")
print("| config | sync | pipelined | block by block | CPU (seq_kernels.c -O2, 1 thread) | pipelined / CPU |")
print("|---|---|---|---|---|---|")
for name, row in d["north_star_pes"]["rows"].items():
    cpu = row.get("cpu")
    print(f"| {name} | {n(row['sync'])} | {n(row['pipelined'])} | {n(row['block_by_block'])} | {n(cpu, 1) if cpu else ''} | "
          f"{n(row['pipelined'] / cpu) + 'x' if cpu else ''} |")
b = d["north_star_pes"]["comb_bank_512"]
print(f"\n512-chain CombPE bank: {b['ms_per_block']:.4f} ms per 48 000-frame block = {n(b['value'], 1)} Msamples/s "
      f"({n(b['chain_msamples_s'])} chain-Msamples/s); CPU {n(b.get('cpu_ms_per_block'), 0)} ms per block ({n(b.get('over_cpu'))}x).")
print("\n### cases\n")
for k, v in d["cases"].items():
    ov = o["cases"].get(k, {})
    print(f"- {k}: {n(v.get('value'), 1)} (round 2: {n(ov.get('value'), 1)}) cpu {v.get('cpu_oracle_msamples_s') or (v.get('cpu_baseline') or {}).get('value')}"
          f" launch {((v.get('roofline') or {}).get('avg_launch_ms') or 0) * 1e3:.2f} us frac {(v.get('roofline') or {}).get('frac')}")
for k in ("voice_mix", "supersaw_mix"):
    v, ov = d[k], o[k]
    print(f"- {k}: {v['ms_per_block']} ms ({v['value']}) round 2 {ov['ms_per_block']} ms; cpu {v['cpu_baseline']['value']} over {v['over_cpu']}")
print("- value", d["value"], d["ms_per_step"], "steps", d["steps"], "round2", o["value"], o["ms_per_step"], o["steps"])
for k in ("roofline", "roofline_one_step", "roofline_scaled", "roofline_filter_alone", "roofline_sine_alone"):
    r = d[k]
    print(f"- {k}: {r['frames_per_launch']} frames {r['avg_launch_ms'] * 1e3:.2f} us {r['achieved']} GB/s frac {r['frac']} traffic {r.get('traffic')}")
w = d["value_with_d2h"]
print("- d2h", w["pipelined"], w["sync"])
print("- cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"])
print("- mfma", d["cases"]["c3_convolve_64k_taps"]["direct_form_mfma"])
