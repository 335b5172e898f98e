#!/usr/bin/env python3
"""profiles/ubench/r04/step_timeline.py DIR [ANCHOR] -- from a rocprofv3 --kernel-trace CSV: the kernels of ONE steady-state
step in start order (offset from the step's first kernel, duration, gap to the previous kernel's end, queue), averaged over
the steps of the middle third of the run.  A step begins at each launch whose name contains ANCHOR (default: the gene-side
update, the first k_update behind a sweep)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("vbnmf::", "")
    return n.split("(")[0][:44]
steps, cur, prev_update = [], [], False
for r in rows:
    name = short(r["Kernel_Name"])
    if "k_update" in name and not prev_update and cur:       # a step starts at the first k_update behind anything else
        steps.append(cur); cur = []
    prev_update = "k_update" in name
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
common = collections.Counter(tuple(k[0] for k in x) for x in steps).most_common(1)[0][0]
steps = [s for s in steps if tuple(k[0] for k in s) == common]
mid = steps[len(steps) // 3: 2 * len(steps) // 3]
print(f"{len(steps)} steps of {len(steps[0])} kernels; averaging {len(mid)}")
n = len(mid[0])
t_end_prev = None
tot = sum(s[-1][2] - s[0][1] for s in mid) / len(mid) / 1e3
period = sum(b[0][1] - a[0][1] for a, b in zip(mid, mid[1:])) / (len(mid) - 1) / 1e3
for i in range(n):
    off = sum(s[i][1] - s[0][1] for s in mid) / len(mid) / 1e3
    dur = sum(s[i][2] - s[i][1] for s in mid) / len(mid) / 1e3
    gap = sum(s[i][1] - max(x[2] for x in s[:i]) for s in mid) / len(mid) / 1e3 if i else 0.0
    print(f"  {mid[0][i][0]:46s} queue {mid[0][i][3]:>3s}  start {off:8.1f} us  dur {dur:7.1f} us  after the latest earlier end {gap:+7.1f} us")
print(f"first start -> last end {tot:.1f} us; step period {period:.1f} us")
