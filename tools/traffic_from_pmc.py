#!/usr/bin/env python3
"""profiles/traffic.json and profiles/<tag>_pmc_traffic.md from the per-(kernel, grid) PMC summaries that
tools/pmc_round.sh left in gpurun_out/ (FETCH_SIZE and WRITE_SIZE, separate rocprofv3 passes, KiB).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a wide (16 B per lane)
coalesced streaming read -> x2 where the kernel reads that way (stated per row); WRITE_SIZE is exact.

    python tools/traffic_from_pmc.py r3
"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"


MAXES = {}


def table(name):
    rows = {}
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_{name}.md")
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) == 7 and cells[1].isdigit():
            rows[(cells[0], int(cells[1]))] = float(cells[4])
            MAXES[(name, cells[0], int(cells[1]))] = float(cells[6])
    return rows


def rows_of(name, kernel):
    """[(grid, mean KiB)] of every row of a kernel (several rows per grid when launches of different sizes share it),
    by ascending mean."""
    found = []
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_{name}.md")
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) == 7 and cells[1].isdigit() and cells[0].startswith(kernel):
            found.append((int(cells[1]), float(cells[4])))
    return sorted(found, key=lambda r: r[1])


def kib(rows, kernel, grid):
    for (k, g), v in rows.items():
        if k.startswith(kernel) and g == grid:
            return v
    raise KeyError((kernel, grid))


out = {"source": f"profiles/{tag}_pmc_traffic.md"}
md = [f"# PMC: HBM traffic of the kernels priced against the HBM roofline, round {tag[1:]}", "",
      f"`bash tools/pmc_round.sh {tag}` on an MI355X box: per probe one `rocprofv3 --pmc FETCH_SIZE --kernel-trace` pass and one "
      "with `WRITE_SIZE` (separate runs, no other trace domain), summarised per (kernel, grid) by `tools/summarize_pmc.py` "
      f"into the `{tag}_pmc_*_raw.md` files next to this one; this file is made from those by `tools/traffic_from_pmc.py`.  "
      "Counter unit = KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a "
      "wide (16 B per lane) coalesced streaming read -> x2 where stated; WRITE_SIZE is exact.", ""]

# ---- C2 fused
sizes = [1_000_000, 16_000_000, 33_000_000, 1 << 26, 134_000_000]       # (134 M: a stream's window since the cap grows with small graphs)
wr_rows = rows_of("c2fused_WRITE_SIZE", "k_biquad_settled<true, true, true")
fe_rows = rows_of("c2fused_FETCH_SIZE", "k_biquad_settled<true, true, true")
assert len(wr_rows) == len(sizes), wr_rows          # one cluster of launches per size, ascending
md += ["## C2, the fused chain: `k_biquad_settled<mono, staged, sine, 256>` (`pgx_biquad_sine`): nothing read, 4 B/frame written", "",
       "| frames per launch | grid (threads) | FETCH_SIZE KiB (largest row of the kernel) | WRITE_SIZE KiB | traffic MB (fetch as counted + write) | algorithmic MB (4 B/frame) | ratio |",
       "|---|---|---|---|---|---|---|"]
out["pgx_biquad_sine"] = {}
fe = max(v for _, v in fe_rows)
for frames, (grid, wr) in zip(sizes, wr_rows):
    total = (fe + wr) * 1024
    out["pgx_biquad_sine"][str(frames)] = int(round(total, -4))
    md.append(f"| {frames:,} | {grid} | {fe:,.1f} | {wr:,.1f} | **{total / 1e6:.2f}** | {4 * frames / 1e6:.2f} | {total / (4 * frames):.3f} |")
md += ["", "(The little that is fetched: coefficient tables and the carried state; what is written beyond the output: the "
       "state.  The kernel is compiled without scratch: no spill traffic.)", ""]

# ---- C2 filter alone
f, w = table("biquad_FETCH_SIZE"), table("biquad_WRITE_SIZE")
md += ["## The filter alone: `k_biquad_settled<mono, staged>` (`pgx_biquad_const`, settle 1024 frames), 8 B/frame", "",
       "| frames per launch | FETCH_SIZE KiB (raw) | read MB (x2: wide streaming reads) | WRITE_SIZE KiB | traffic MB | algorithmic MB | ratio |",
       "|---|---|---|---|---|---|---|"]
out["pgx_biquad_const_settled"] = {}
grids = {1_000_000: 126976, 16_000_000: 253952, 33_000_000: 258048, 1 << 26: 262144}      # 512 workgroups of 512 threads
for frames, grid in grids.items():
    fe, wr = kib(f, "k_biquad_settled<true, true, false", grid), kib(w, "k_biquad_settled<true, true, false", grid)
    total = (2 * fe + wr) * 1024
    out["pgx_biquad_const_settled"][str(frames)] = int(round(total, -4))
    md.append(f"| {frames:,} | {fe:,.1f} | {2 * fe * 1024 / 1e6:.2f} | {wr:,.1f} | **{total / 1e6:.2f}** | {8 * frames / 1e6:.2f} | {total / (8 * frames):.3f} |")
md.append("")

# ---- C3
f, w = table("c3_FETCH_SIZE"), table("c3_WRITE_SIZE")
shapes = [("96 000 frames, stereo (2 packed transforms)", 96000, 65536), ("65 537 frames, stereo (one shared transform)", 65537, 32768),
          ("1 440 000 frames (22 packed transforms)", 1440000, 720896)]
if tag >= "r4":        # renders of three hops and more take the 2^18-point transform (convolve_pe.device_fft_size)
    shapes[2] = ("1 440 000 frames (8 blocks of 196 609 frames, N = 262 144: 4 packed transforms)", 1440000, 262144)
md += ["## C3: `pgx_convolve_fft`, 65 536 taps, N = 131 072 (three launches per call)", "",
       "| shape | FETCH_SIZE KiB (three kernels, raw) | WRITE_SIZE KiB | traffic MB (reads x2: upper bound) | algorithmic MB (16 B/frame) | ratio |",
       "|---|---|---|---|---|---|"]
out["pgx_convolve_fft"] = {}
for label, frames, grid in shapes:
    fe = kib(f, "k_fft_cols<0", grid) + kib(f, "k_fft_rows<true", grid) + kib(f, "k_fft_cols<2", grid)
    wr = kib(w, "k_fft_cols<0", grid) + kib(w, "k_fft_rows<true", grid) + kib(w, "k_fft_cols<2", grid)
    total = (2 * fe + wr) * 1024
    out["pgx_convolve_fft"][str(frames)] = int(round(total, -4))
    md.append(f"| {label} | {fe:,.1f} | {wr:,.1f} | **{total / 1e6:.1f}** | {16 * frames / 1e6:.2f} | {total / (16 * frames):.1f} |")
md += ["", "The float64 work buffer crosses HBM three times (forward columns -> rows x spectrum -> inverse columns); "
       "the three passes are what the four-step transform needs with 1024-point tiles (DESIGN.md section 7).", ""]

# ---- CombPE
f, w = table("comb_FETCH_SIZE"), table("comb_WRITE_SIZE")
md += ["## CombPE: `k_comb_poly` (`pgx_comb`, scalar frequency, mono, 440 Hz / D = 100)", "",
       "| launch | kernels | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB (as counted) | algorithmic MB (8 B per chain frame) |", "|---|---|---|---|---|---|"]
out["pgx_comb"] = {}
for label, key, frames, parts in (("one 44 100-frame block (one segment: the reference's loop)", "44100", 44100, [("k_comb_poly<1", 256)]),
                                  ("a look-ahead window of 64 blocks, 2 822 400 frames (441 time segments)", "2822400", 2822400,
                                   [("k_comb_poly<0", 44288), ("k_comb_poly<1", 44288)]),
                                  ("512-chain bank x 48 000 frames", "bank512x48000", 512 * 48000,
                                   [("k_comb_poly<0", 1310720), ("k_comb_poly<1", 1310720)])):
    fe = sum(kib(f, k, g) for k, g in parts)
    wr = sum(kib(w, k, g) for k, g in parts)
    total = (fe + wr) * 1024
    out["pgx_comb"][key] = int(round(total, -3))
    md.append(f"| {label} | {' + '.join(k + '>' for k, _ in parts)} | {fe:,.1f} | {wr:,.1f} | **{total / 1e6:.2f}** | {8 * frames / 1e6:.2f} |")
md += ["", "(A lane's loads are 4-byte accesses D frames apart -- not the wide streaming reads the x2 correction is for: "
       "FETCH_SIZE as counted.  The segmented renders read the input twice (reduce + apply): 12 B per frame.)", ""]

# ---- the mixes
f, w = table("mixes_FETCH_SIZE"), table("mixes_WRITE_SIZE")
md += ["## The sharded mixes at one GPU (per 48 000-frame block)", "",
       "| kernel | grid | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB |", "|---|---|---|---|"]
for (k, g), v in sorted(f.items()):
    if k.startswith("__amd") or k.startswith("k_fill") or k.startswith("k_copy"):
        continue
    md.append(f"| `{k}` | {g} | {v:,.1f} | {w.get((k, g), 0.0):,.1f} |")
# (k_mix_batch runs at two sizes in the probe -- 512 inputs for the SuperSaw mix, 64 for C4 -- its largest launch is the 512-input one)
mix512 = max(v for (n_, k, g), v in MAXES.items() if n_ == "mixes_FETCH_SIZE" and k.startswith("k_mix_batch"))
ss = (kib(w, "k_supersaw_wide<4>", 131072) + 2 * mix512) * 1024
on_chip = any(k.startswith("k_voice_tiles") for (k, g) in f)
if on_chip:
    # round 4, late: C5's voices are mixed on chip (pgx_voice_tiles).  What still crosses HBM per block: the envelopes (written by
    # the walk, read once by k_voice_tiles: wide reads, x2), the groups' rows of partial sums (written, read once by
    # k_mix_partials: x2), the tiles' entries, the final mix
    vt = [(k, g) for (k, g) in f if k.startswith("k_voice_tiles")]
    vt_key = max(vt, key=lambda kg: kg[1])
    c5 = (2 * f[vt_key] + w.get(vt_key, 0.0) + kib(w, "k_adsr_walk_par<8>", 262144) + 2 * kib(f, "k_mix_partials", 48128)
          + kib(w, "k_mix_partials", 48128) + kib(f, "k_voice_tile_entries", 131072) + kib(w, "k_voice_tile_entries", 131072)) * 1024
    c5_text = (f"C5 (voices mixed on chip, `pgx_voice_tiles`): the `[512][48000]` layer of voices is gone; what crosses HBM is the envelope "
               f"layer (float32, written by the walk and read once by `k_voice_tiles`), the groups' rows of partial sums and the tiles' "
               f"entries: about **{c5 / 1e6:.0f} MB** per block (394 MB with the layered path, `r3_pmc_traffic.md`).")
else:
    c5 = (kib(w, "k_blitsaw_biquad_wide", 131072) + kib(w, "k_adsr_walk_par<8>", 262144) + 2 * kib(f, "k_gain_mix_batch", 48128)) * 1024
    c5_text = (f"C5: oscillator+filter output and envelopes, each `[512][48000]`, written and read by "
               f"`k_gain_mix_batch`: about **{c5 / 1e6:.0f} MB** per block.")
md += ["", f"SuperSaw mix (512 x 7 oscillators): the `[512][48000]` float32 layer under the MixPE is written once and read once "
       f"(x2 on the wide reads of `k_mix_batch`; the 512-input launch is the table's maximum, its mean blends in C4's 64-input mix): about **{ss / 1e6:.0f} MB** per block against "
       f"0.192 MB of final mix.  {c5_text}  See DESIGN.md section 7.", ""]
out["supersaw_mix_512"] = {"48000": int(round(ss, -5))}
out["c5_voice_mix_512"] = {"48000": int(round(c5, -5))}

# keep the round-2 entries that were not measured again
try:
    old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for k, v in old.items():
        if k not in out:
            out[k] = v
except OSError:
    pass
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"))
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.md"), "w").write("\n".join(md) + "\n")
for name in ("c2fused", "biquad", "c3", "comb", "mixes"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_{name}_{c}.md")
        open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{name}_{c}_raw.md"), "w").write(open(src).read())
print(json.dumps(out)[:600])
