#!/usr/bin/env python3
"""profiles/summarize_pmc.py TAG -- fold the rocprofv3 outputs of profiles/collect.sh into small files:
gpurun_out/TAG_summary/{TAG_kernel_stats.csv, TAG_pmc_<pass>.csv (per-kernel means), TAG_traffic.json}.
HBM bytes per k_sweep launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; the doubling of FETCH_SIZE is
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section: it holds for WIDE COALESCED
STREAMING reads, 16 B per lane -- the sweep's entry stream).  k_update gathers 8-byte elements row by row: the guide calls
other access widths uncalibrated, so its bytes are reported AS COUNTED (FETCH_SIZE + WRITE_SIZE), with the doubled figure
beside it for reference (the counted 49.9 MB already equal the kernel's expected reads at rank 10)."""
import csv, glob, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = os.path.join("gpurun_out", f"{tag}_summary")
os.makedirs(out, exist_ok=True)

def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None

def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("vbnmf::", "").strip()

stats = find(f"gpurun_out/{tag}_stats/**/*kernel_stats.csv")
if stats:
    import shutil
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))

means = {}
for d in sorted(glob.glob(f"gpurun_out/{tag}_pmc_*")):
    if not os.path.isdir(d):
        continue
    cc = find(d + "/**/*counter_collection.csv")
    if not cc:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        for key in ("k_sweep1", "k_sweep", "k_ml_update", "k_ml_final", "k_update2", "k_update", "k_final", "k_prime", "k_control", "k_pack", "k_tail"):
            if key in r["Kernel_Name"]:
                k = key
                break
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    name = os.path.basename(d)[len(tag) + 1:]
    with open(os.path.join(out, f"{tag}_{name}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
        for k in sorted(acc):
            for c in sorted(acc[k]):
                v = acc[k][c]
                w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
                means[(k, c)] = sum(v) / len(v)

fs, ws = means.get(("k_sweep", "FETCH_SIZE")), means.get(("k_sweep", "WRITE_SIZE"))
traffic = {"kernel": "k_sweep", "FETCH_SIZE_KB_per_launch": fs, "WRITE_SIZE_KB_per_launch": ws,
           "sweep_hbm_bytes_per_launch": (2 * fs + ws) * 1024 if fs and ws else None,
           "sweep_hbm_bytes_uncorrected": (fs + ws) * 1024 if fs and ws else None,
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) over bench.py; FETCH_SIZE doubled as the gfx950 "
                   "note in MI355X_MICROARCH.md prescribes; units KB",
           "k_update_FETCH_SIZE_KB": means.get(("k_update", "FETCH_SIZE")),
           "k_update_WRITE_SIZE_KB": means.get(("k_update", "WRITE_SIZE")),
           "k_update_hbm_bytes_per_launch_as_counted": ((means.get(("k_update", "FETCH_SIZE")) or 0) + (means.get(("k_update", "WRITE_SIZE")) or 0)) * 1024 or None,
           "k_update_hbm_bytes_if_fetch_doubled": (2 * (means.get(("k_update", "FETCH_SIZE")) or 0) + (means.get(("k_update", "WRITE_SIZE")) or 0)) * 1024 or None,
           "sq_counters_k_sweep": {c: v for (k, c), v in means.items() if k == "k_sweep" and c.startswith("SQ_")},
           "l2_counters_k_sweep": {c: v for (k, c), v in means.items() if k == "k_sweep" and (c.startswith("TCC") or c.startswith("GRBM"))}}
json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "sweep_hbm_bytes_per_launch")}))
