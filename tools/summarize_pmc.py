#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE CSVs of tools/round_evidence.sh into per-kernel HBM traffic figures (bytes per launch):
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KiB; per MI355X_MICROARCH.md (HBM / rocprofv3 section) FETCH_SIZE
on gfx950 counts half the bytes of wide coalesced reads -> x2, WRITE_SIZE is used as is.  First launch of every kernel is skipped.
Usage: summarize_pmc.py OUTDIR  -> JSON on stdout"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short_name(kernel):
    """_ZN2fa16fa_fwd_pp_kernelI... -> fa_fwd_pp_kernel; demangled fa::name<...> -> name; anything else unchanged"""
    m = re.match(r"_ZN2fa(\d+)", kernel)
    if m:
        n = int(m.group(1))
        return kernel[m.end():m.end() + n]
    m = re.search(r"fa::(\w+)", kernel)
    return m.group(1) if m else kernel


def per_kernel(directory):
    """{kernel short name: mean counter value per launch (summed over counter instances), launches}"""
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    per_dispatch = defaultdict(float)
    name_of = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            did = row.get("Dispatch_Id") or row.get("Dispatch_ID")
            per_dispatch[did] += float(row["Counter_Value"])
            name_of[did] = row["Kernel_Name"]
    by_kernel = defaultdict(list)
    for did in sorted(per_dispatch, key=lambda x: int(x)):
        by_kernel[short_name(name_of[did])].append(per_dispatch[did])
    return {k: {"mean": sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0], "launches": len(v)} for k, v in by_kernel.items()}


def main():
    out = sys.argv[1]
    res = {}
    for tag in ("fwd_c3", "bwd_c4"):
        fetch, write = per_kernel(os.path.join(out, tag + "_FETCH_SIZE")), per_kernel(os.path.join(out, tag + "_WRITE_SIZE"))
        for k in fetch:
            if not k.startswith("fa_"):
                continue
            rd = fetch[k]["mean"] * 1024 * 2
            wr = write.get(k, {"mean": 0.0})["mean"] * 1024
            res[f"{tag}:{k}"] = {"FETCH_SIZE_KiB_raw": fetch[k]["mean"], "WRITE_SIZE_KiB_raw": write.get(k, {"mean": 0.0})["mean"],
                                 "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
                                 "launches_profiled": fetch[k]["launches"]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
