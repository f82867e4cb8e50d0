#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, average / min /
max duration, registers and LDS.  Grid size separates the launches of one kernel at different
problem sizes (e.g. the 1M-frame and the 2^26-frame biquad launches of bench.py), which the
stock --stats summary averages together.

    python tools/summarize_trace.py gpurun_out/prof/<pid>_kernel_trace.csv > profiles/<name>.md
"""

import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name)


def main(path):
    groups = defaultdict(list)
    meta = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]), int(row["Grid_Size_Y"]),
                   int(row["Grid_Size_Z"]), int(row["Workgroup_Size_X"]))
            groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"])
    total = sum(sum(v) for v in groups.values())
    print("| kernel | grid (threads) | wg | calls | avg us | min us | max us | total ms | % | vgpr | agpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        name, gx, gy, gz, wg = key
        vg, ag, sg, lds = meta[key]
        print(f"| {name} | {gx}x{gy}x{gz} | {wg} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | "
              f"{min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {sum(v) / 1e6:.3f} | {100.0 * sum(v) / total:.1f} | "
              f"{vg} | {ag} | {sg} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])
