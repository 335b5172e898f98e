#!/usr/bin/env python3
"""profiles/ubench/r05/kernel_durations.py DIR NAME... -- mean / min / max duration of the kernels whose name contains NAME, from a
rocprofv3 --kernel-trace CSV (the first fifth of the launches dropped)."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for key in sys.argv[2:]:
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
    v = v[len(v) // 5:]
    if v:
        print(f"   {key:12s} launches {len(v):5d}  mean {sum(v) / len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}")
