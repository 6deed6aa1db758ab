#!/usr/bin/env python3
"""gpurun_out/evidence (tools/round_evidence.sh) -> profiles/rNN_*: every summary carries the library's source digest (fa_build_info
`src=`) and the git commit the tree was built from, so that a table can be tied to the kernels that produced it.
Usage: round_evidence_collect.py [EVIDENCE_DIR] [ROUND_TAG]"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import per_kernel, short_name  # noqa: E402
import summarize_counters  # noqa: E402

EV = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "evidence")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r4"
PROF = os.path.join(ROOT, "profiles")


def line(path):
    with open(path) as f:
        for ln in f:
            if ln.startswith("{"):
                return json.loads(ln)
    raise SystemExit(f"no JSON line in {path}")


def stats(directory):
    """kernel short name -> (calls, avg ns, total ns) from rocprofv3's kernel_stats.csv; also returns the file path"""
    files = glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel_stats.csv under {directory}")
    if len(files) > 1:
        raise SystemExit(f"{len(files)} kernel_stats.csv under {directory}: stale files of an earlier run were merged into it - `rm -rf` the evidence "
                         "directory on the dev side before tools/round_evidence.sh (gpurun merges, it does not mirror)")
    out = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            out[short_name(row["Name"])] = dict(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]), total_ns=float(row["TotalDurationNs"]),
                                                min_ns=float(row["MinNs"]), max_ns=float(row["MaxNs"]), name=row["Name"])
    return out, files[0]


def main():
    b = line(os.path.join(EV, "bench_stdout.json"))
    bp = line(os.path.join(EV, "bench_stdout_under_rocprof.json"))
    b4 = line(os.path.join(EV, "bench_c4_stdout_under_rocprof.json"))
    digest = b["roofline"]["library_source_digest"]
    commit = b.get("git_commit")
    for other in (bp, b4):
        assert other["roofline"]["library_source_digest"] == digest
    tagline = {"library": b["library"], "library_source_digest": digest, "git_commit": commit, "device": b["device"]}
    shutil.copy(os.path.join(EV, "bench_stdout.json"), os.path.join(PROF, f"{TAG}_bench_stdout.json"))
    shutil.copy(os.path.join(EV, "bench_stdout_under_rocprof.json"), os.path.join(PROF, f"{TAG}_bench_stdout_under_rocprof.json"))
    shutil.copy(os.path.join(EV, "bench_c4_stdout_under_rocprof.json"), os.path.join(PROF, f"{TAG}_bench_c4_stdout_under_rocprof.json"))
    st3, f3 = stats(os.path.join(EV, "bench_stats"))
    st4, f4 = stats(os.path.join(EV, "bench_c4_stats"))
    shutil.copy(f3, os.path.join(PROF, f"{TAG}_bench_kernel_stats.csv"))
    shutil.copy(f4, os.path.join(PROF, f"{TAG}_bench_c4_kernel_stats.csv"))
    fwd_name = b["roofline"]["kernel"]                       # the kernel bench.py timed (fa_fwd_kernel_name)
    fwd = st3[fwd_name]
    flops = b["roofline"]["algorithmic_flops_per_launch"]
    frac_prof = flops / (fwd["avg_ns"] * 1e-9) / 1e12 / b["roofline"]["peak"]
    # HBM traffic (FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md, WRITE_SIZE as is; first launch of each kernel skipped)
    hbm = {}
    for tag in ("fwd_c3", "bwd_c4"):
        fe, wr = per_kernel(os.path.join(EV, tag + "_FETCH_SIZE")), per_kernel(os.path.join(EV, tag + "_WRITE_SIZE"))
        for k in fe:
            if k.startswith("fa_"):
                rd, w = fe[k]["mean"] * 1024 * 2, wr.get(k, {"mean": 0.0})["mean"] * 1024
                hbm[f"{tag}:{k}"] = {"read_bytes_corrected": rd, "write_bytes": w, "traffic_bytes_per_launch": rd + w, "launches_profiled": fe[k]["launches"]}
    f = hbm["fwd_c3:" + fwd_name]
    with open(os.path.join(PROF, f"{TAG}_hbm_traffic.json"), "w") as fo:
        json.dump({"workload": "c3", "kernel": fwd_name, "traffic_bytes_per_launch": f["traffic_bytes_per_launch"],
                   "read_bytes_corrected": f["read_bytes_corrected"], "write_bytes": f["write_bytes"],
                   "algorithmic_bytes_per_launch": 2155872256, **tagline,
                   "source": "tools/round_evidence.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per "
                             "MI355X_MICROARCH.md, WRITE_SIZE as is; mean of launches 2-4 of tools/run_fwd_once.py --seq 16384 --causal 1"}, fo, indent=1)
    # shader counters
    sq = {}
    for prefix in sorted({re.sub(r"_g\d+$", "", p) for p in glob.glob(os.path.join(EV, "sq", "*_g*")) if os.path.isdir(p)}):
        for kname, k in summarize_counters.read_case(prefix).items():
            sq[f"{os.path.basename(prefix)}:{kname}"] = {"counters": k["counters"], "derived": summarize_counters.derive(k)}
    with open(os.path.join(PROF, f"{TAG}_shader_pmc_summary.json"), "w") as fo:
        json.dump({**tagline, "kernels": sq}, fo, indent=1)
    c4 = {k: v for k, v in st4.items() if k.startswith("fa_")}
    summary = {
        **tagline,
        "what": "one MI355X box, one lease, in this order: plain bench.py; rocprofv3 --kernel-trace --stats of bench.py (c3) and of bench.py --workload c4; "
                "FETCH_SIZE / WRITE_SIZE passes; three SQ counter passes (tools/round_evidence.sh)",
        "bench_line": {k: b[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "frac_of_fp16_mfma_peak")},
        "roofline_from_bench_hip_events": {k: b["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launch_ms_distribution", "power_and_sclk")},
        "roofline_from_rocprof_stats": {"file": f"profiles/{TAG}_bench_kernel_stats.csv", "kernel": fwd["name"][:80], "calls": fwd["calls"], "avg_ms": fwd["avg_ns"] / 1e6,
                                        "min_ms": fwd["min_ns"] / 1e6, "achieved_tflops": flops / (fwd["avg_ns"] * 1e-9) / 1e12, "frac": frac_prof,
                                        "bench_under_rocprof_frac": bp["roofline"]["frac"], "bench_under_rocprof_avg_launch_ms": bp["roofline"]["avg_launch_ms"]},
        "frac_bench_vs_rocprof_relative_difference": abs(frac_prof - b["roofline"]["frac"]) / b["roofline"]["frac"],
        "c4_kernels_from_rocprof_stats": {k: {"calls": v["calls"], "avg_ms": v["avg_ns"] / 1e6} for k, v in c4.items()},
        "c4_bench_line_under_rocprof": {k: b4[k] for k in ("value", "ms_per_step")},
        "hbm_traffic": hbm,
    }
    with open(os.path.join(PROF, f"{TAG}_evidence_summary.json"), "w") as fo:
        json.dump(summary, fo, indent=1)
    print(json.dumps({k: summary[k] for k in ("library_source_digest", "git_commit", "bench_line", "roofline_from_rocprof_stats", "frac_bench_vs_rocprof_relative_difference",
                                              "c4_kernels_from_rocprof_stats")}, indent=1))
    print("fwd c3 traffic / algorithmic:", f["traffic_bytes_per_launch"] / 2155872256)


if __name__ == "__main__":
    main()
