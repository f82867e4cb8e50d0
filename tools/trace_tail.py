#!/usr/bin/env python3
"""The last N kernels of a rocprofv3 kernel trace (gpurun_out/<name>/...kernel_trace.csv) as a timeline: start, end,
duration in us from the first of them, queue, kernel."""
import csv, glob, sys
name, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(f"gpurun_out/{name}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
base = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - base) / 1000, (int(r["End_Timestamp"]) - base) / 1000
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
