#!/usr/bin/env python3
"""profiles/ubench/r04/update_split.py DIR -- from a rocprofv3 --kernel-trace CSV of bench.py: mean duration of k_update by
side.  In every step the gene-side launch (W) comes first, the cell-side one (H) second, so within the stream's dispatch
order the k_update launches alternate W, H, W, H ...; k_sweep and the rest are reported as they are."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
flip = 0
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_update" in name:
        acc["k_update(W)" if flip == 0 else "k_update(H)"].append(dur)
        flip ^= 1
    elif "k_sweep" in name:
        acc["k_sweep"].append(dur)
        flip = 0                      # a sweep closes the step: the next update is a gene-side one
    elif "k_final" in name:
        acc["k_final"].append(dur)
    elif "k_prime" in name:
        flip = 0
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]               # drop the ramp
    print(f"{k:14s} launches {len(v):6d}  mean {sum(v) / len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}")
