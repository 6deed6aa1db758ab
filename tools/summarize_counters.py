#!/usr/bin/env python3
"""Reduce the shader-counter CSVs of tools/round_evidence.sh: per case and kernel, mean counter value per launch (summed over counter instances, first
launch skipped), kernel time under the profiler, and a few derived figures: effective clock (GRBM_GUI_ACTIVE / 8 XCDs / time), MFMA
pipe utilisation at that clock (SQ_VALU_MFMA_BUSY_CYCLES / 4 SIMDs per CU / 256 CUs / cycles), wave-cycle breakdown, LDS-array
utilisation (SQ_LDS_IDX_ACTIVE / 256 CUs / cycles).  Usage: summarize_counters.py OUTDIR -> JSON on stdout"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

from summarize_pmc import short_name


def read_case(prefix):
    out = defaultdict(lambda: {"counters": {}, "ms": []})
    for d in sorted(glob.glob(prefix + "_g*")):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        per = defaultdict(lambda: defaultdict(float))
        meta = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                did = int(row["Dispatch_Id"])
                per[did][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[did] = (short_name(row["Kernel_Name"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        by_kernel = defaultdict(list)
        for did in sorted(per):
            by_kernel[meta[did][0]].append(did)
        for k, dids in by_kernel.items():
            if not k.startswith("fa_"):
                continue
            use = dids[1:] if len(dids) > 1 else dids
            for cname in per[use[0]]:
                out[k]["counters"][cname] = sum(per[d_][cname] for d_ in use) / len(use)
            out[k]["ms"].append(sum(meta[d_][1] for d_ in use) / len(use))
    return out


def derive(k):
    c, ms = k["counters"], sum(k["ms"]) / len(k["ms"])
    d = {"kernel_ms_under_pmc": ms}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d["effective_clock_GHz"] = cyc / (ms * 1e6)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_pipe_utilisation_at_effective_clock"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / 256.0 / cyc
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_array_utilisation"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        w = c["SQ_WAVE_CYCLES"]
        d["wave_cycles_breakdown"] = {"SQ_WAIT_ANY (parked: s_waitcnt / barrier)": c["SQ_WAIT_ANY"] / w,
                                      "SQ_WAIT_INST_ANY (issue stall)": c.get("SQ_WAIT_INST_ANY", 0.0) / w,
                                      "SQ_ACTIVE_INST_ANY": c.get("SQ_ACTIVE_INST_ANY", 0.0) / w}
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_INSTS_LDS"):
        d["valu_insts_per_lds_inst"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_LDS"]
    if c.get("SQ_INSTS_VMEM_RD") and c.get("SQ_INSTS_LDS"):
        # wave-level vector-memory loads (LDS-DMA pieces, statistics, Q / dO / O rows) against LDS instructions: a per-tile figure follows from the kernel's known
        # LDS instructions per wave and tile (the vector-memory issue cost of dK/dV)
        d["vmem_rd_insts_per_lds_inst"] = c["SQ_INSTS_VMEM_RD"] / c["SQ_INSTS_LDS"]
        d["vmem_wr_insts_per_lds_inst"] = c.get("SQ_INSTS_VMEM_WR", 0.0) / c["SQ_INSTS_LDS"]
    return d


def main():
    root = sys.argv[1]
    res = {}
    for prefix in sorted({re.sub(r"_g\d+$", "", p) for p in glob.glob(os.path.join(root, "*_g*")) if os.path.isdir(p)}):
        case = os.path.basename(prefix)
        for kname, k in read_case(prefix).items():
            res[f"{case}:{kname}"] = {"counters": k["counters"], "derived": derive(k)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
