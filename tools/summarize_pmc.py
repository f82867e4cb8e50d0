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
        print(f"| {k} | {g} | {c} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
