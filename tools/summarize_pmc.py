#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per (kernel, grid size, counter).

    python tools/summarize_pmc.py <..._counter_collection.csv>

FETCH_SIZE / WRITE_SIZE are reported by the tool in KiB; the gfx950 x2 correction for wide
streaming reads (MI355X_MICROARCH.md, HBM section) is NOT applied here -- the profiles/*.md
files that quote bytes state it explicitly."""

import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name)


def main(path):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(short(row["Kernel_Name"]), int(row["Grid_Size"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("| kernel | grid (threads) | counter | launches | mean | min | max |")
    print("|---|---|---|---|---|---|---|")
    for (k, g, c), v in sorted(acc.items()):
        # launches of different sizes can share a grid (a fixed number of workgroups): one row per cluster of values
        # (a value more than 20 % above its cluster's smallest starts a new one)
        clusters = []
        for x in sorted(v):
            if clusters and x <= 1.2 * clusters[-1][0] + 1.0:
                clusters[-1].append(x)
            else:
                clusters.append([x])
        for cl in clusters:
            print(f"| {k} | {g} | {c} | {len(cl)} | {sum(cl) / len(cl):.1f} | {min(cl):.1f} | {max(cl):.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
