#!/usr/bin/env python3
"""Print the last N kernel launches of a rocprofv3 kernel trace as a timeline (start offset, duration, gap, stream)."""
import csv, sys
path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end, e)
