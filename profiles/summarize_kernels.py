#!/usr/bin/env python3
"""profiles/summarize_kernels.py TAG NAME [NAME ...] -- fold rocprofv3 outputs gpurun_out/TAG_NAME_stats/ and
gpurun_out/TAG_NAME_pmc_{fetch_size,write_size}/ (profiles/collect_r04_evidence.sh) into
gpurun_out/TAG_summary/{TAG_NAME_kernel_stats.csv, TAG_NAME_traffic.json}.

HBM bytes per launch and kernel.  /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half the bytes of a WIDE COALESCED STREAMING read (16 B per lane) -- that doubling is applied to the sweep kernels only
(k_sweep, k_sweep1, k_spmm: their traffic is the packed entry stream read with dwordx4 loads per lane).  The update /
pack kernels gather 8-byte elements row by row: other access widths are uncalibrated per the guide, so their FETCH_SIZE
is reported as counted (`hbm_bytes_as_counted`) with the doubled figure beside it (`hbm_bytes_if_fetch_doubled`) --
VERDICT r03 weak #3: 49.9 MB counted already equals k_update's expected reads at rank 10."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

STREAMING = ("k_sweep1", "k_sweep", "k_spmm")
KEYS = ("k_sweep1", "k_sweep", "k_spmm", "k_ml_update", "k_ml_final", "k_ml_control", "k_update2", "k_update", "k_final", "k_prime",
        "k_control", "k_pack", "k_tail_h", "k_tail_data", "k_tail", "k_group_sum", "k_gamma_init", "k_ctl_init")


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def kernel_key(name):
    for key in KEYS:
        if key in name:
            return key
    return name.split("(")[0].replace("void ", "").strip()


def main():
    tag, names = sys.argv[1], sys.argv[2:]
    out = os.path.join("gpurun_out", f"{tag}_summary")
    os.makedirs(out, exist_ok=True)
    for name in names:
        stats = find(f"gpurun_out/{tag}_{name}_stats/**/*kernel_stats.csv") or find(f"gpurun_out/{tag}_{name}_auto_stats/**/*kernel_stats.csv")
        if stats:
            shutil.copy(stats, os.path.join(out, f"{tag}_{name}_kernel_stats.csv"))
        means = {}
        for counter in ("fetch_size", "write_size"):
            cc = find(f"gpurun_out/{tag}_{name}_pmc_{counter}/**/*counter_collection.csv")
            if not cc:
                continue
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(cc)):
                acc[(kernel_key(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in acc.items():
                means[(k, c)] = (sum(v) / len(v), len(v))
        kernels = sorted({k for k, _ in means})
        table = {}
        for k in kernels:
            f = means.get((k, "FETCH_SIZE"), (None, 0))
            w = means.get((k, "WRITE_SIZE"), (None, 0))
            if f[0] is None or w[0] is None:
                continue
            counted = (f[0] + w[0]) * 1024
            doubled = (2 * f[0] + w[0]) * 1024
            table[k] = {"dispatches": f[1], "FETCH_SIZE_KB": f[0], "WRITE_SIZE_KB": w[0],
                        "hbm_bytes_as_counted": counted, "hbm_bytes_if_fetch_doubled": doubled,
                        "fetch_correction_applied": k in STREAMING,
                        "hbm_bytes_per_launch": doubled if k in STREAMING else counted}
        json.dump({"tag": tag, "run": name, "note": __doc__.split("\n\n", 1)[1], "kernels": table},
                  open(os.path.join(out, f"{tag}_{name}_traffic.json"), "w"), indent=1)
        print(name, {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in table.items()})


if __name__ == "__main__":
    main()
