#!/usr/bin/env python3
"""profiles/ubench/r05/kernel_means.py DIR -- from a rocprofv3 --kernel-trace CSV of bench.py: mean duration per kernel
(k_update2 = both posterior updates in one launch; k_update by side when the two-launch form ran: W first, H second), and
the mean GAP between consecutive kernels of a step (end of one to start of the next)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
flip = 0
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur = (t1 - t0) / 1e3
    key = None
    if "k_update2" in name:
        key = "k_update2"
    elif "k_update" in name:
        key = "k_update(W)" if flip == 0 else "k_update(H)"
        flip ^= 1
    elif "k_sweep" in name:
        key = "k_sweep"; flip = 0
    elif "k_final" in name:
        key = "k_final"
    elif "k_prime" in name:
        flip = 0
    if key:
        acc[key].append(dur)
        if prev_end is not None and (t0 - prev_end) < 50000:
            acc["gap before " + key].append((t0 - prev_end) / 1e3)
    prev_end = t1
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]               # drop the ramp
    print(f"{k:24s} launches {len(v):6d}  mean {sum(v) / len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}")
