#!/usr/bin/env python3
"""MI355X counterpart of the reference's benchmark.sh + utils/plot_kernels.py:
per sweep point (b=4, h=16 as in benchmark.sh:17-23, d in {64,128}, causal in {F,T}, 12 seqlens) and per kernel - forward, dQ, dK/dV
(the reference profiles its backward kernels too, utils/plot_kernels.py:230-261) - the kernel duration and the matrix-pipe busy fraction (their `sm__throughput.avg.pct_of_peak_sustained_
elapsed`), from ONE rocprofv3 --kernel-trace --pmc pass, as a CSV + an ASCII bar chart.

    python tools/sweep_profile.py                 # spawns rocprofv3 around `--run`, parses, prints, writes CSV
    python tools/sweep_profile.py --run           # the profiled workload itself (one launch per point)

MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): matrix-pipe cycles
over elapsed shader cycles at the clock the kernel actually ran at; the TFLOP/s column uses wall time.
"""
import argparse
import csv
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQS = (512, 1024, 2048, 4096, 8192, 16384, 500, 1000, 2000, 4000, 8000, 16000)     # benchmark.sh:19-21
B, H = 4, 16


def points():
    for d in (128, 64):
        for causal in (0, 1):
            for s in SEQS:
                yield d, causal, s


KERNELS = (("fa_fwd_pp", "fwd", 4.0), ("fa_bwd_dq", "dq", 6.0), ("fa_bwd_dkdv", "dkdv", 8.0))   # name prefix (fa_fwd_pp_kernel / fa_fwd_pp16_kernel by problem size), tag, EXECUTED flop multiple of b*h*sq*sk*d


def run_workload():
    sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
    import torch
    from flash_attn_turing import capi

    dev = torch.device("cuda:0")
    for d, causal, s in points():
        gen = torch.Generator(device=dev).manual_seed(s)
        q, k, v, do = (torch.randn(B, s, H, d, device=dev, dtype=torch.float16, generator=gen) for _ in range(4))
        o, lse = torch.empty_like(q), torch.empty(B, H, s, device=dev, dtype=torch.float32)
        dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(lse)
        for _ in range(2):                                   # first pass warms clocks and caches, the second one is read
            capi.mha_fwd(q, k, v, o, lse, bool(causal))
            capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, bool(causal))      # two launches: dQ (computes D), dK/dV
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_profile.csv"))
    a = ap.parse_args()
    if a.run:
        run_workload()
        return
    tmp = tempfile.mkdtemp(prefix="fa_sweep_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                           "--output-format", "csv", "-d", tmp, "-o", "sweep", "--", sys.executable, os.path.abspath(__file__), "--run"],
                          cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for kname, tag, mult in KERNELS:
        disp, dur = {}, {}
        for r in csv.DictReader(open(os.path.join(tmp, "sweep_counter_collection.csv"))):
            if kname in r["Kernel_Name"]:
                disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = disp.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for r in csv.DictReader(open(os.path.join(tmp, "sweep_kernel_trace.csv"))):
            if kname in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        ids = sorted(disp)[1::2]                          # second launch of every point
        assert len(ids) == len(list(points())), (kname, len(ids))
        for (d, causal, s), i in zip(points(), ids):
            c = disp[i]
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
            flops = mult * B * H * s * s * d * (0.5 if causal else 1.0)
            rows.append(dict(kernel=tag, d=d, causal=causal, seq=s, ms=dur[i], executed_tflops=flops / dur[i] / 1e9, mfma_busy_pct=100 * busy,
                             clock_ghz=c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[i] * 1e6)))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        bar = "#" * int(r["mfma_busy_pct"] / 2)
        print(f"{r['kernel']:5s} d={r['d']:3d} causal={r['causal']} seq={r['seq']:6d} {r['ms']:8.3f} ms {r['executed_tflops']:7.1f} TF  clk {r['clock_ghz']:.2f} GHz  "
              f"MFMA busy {r['mfma_busy_pct']:5.1f}% {bar}")
    print("written", a.out)
    try:
        plot(rows, a.out[:-4])
    except Exception as e:  # noqa: BLE001 - the CSV is the product, the charts are a convenience
        print("plot failed:", e)


def plot(rows, prefix):
    """two charts in the spirit of the reference's utils/plot_kernels.py (:230-261 draws the backward kernels too): executed TFLOP/s and
    MFMA-busy % per sequence length, one line per kernel, d = 128 and d = 64, non-causal and causal"""
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    for key, ylabel, out in (("executed_tflops", "TFLOP/s of executed MFMA work", "_tflops.png"), ("mfma_busy_pct", "MFMA pipes busy, % of elapsed shader cycles", "_mfma_busy.png")):
        fig, axes = plt.subplots(2, 2, figsize=(11, 7), sharey=True)
        for ax, (d, causal) in zip(axes.flat, ((128, 0), (128, 1), (64, 0), (64, 1))):
            for tag in ("fwd", "dq", "dkdv"):
                pts = sorted((r["seq"], r[key]) for r in rows if r["kernel"] == tag and r["d"] == d and r["causal"] == causal)
                ax.plot([p[0] for p in pts], [p[1] for p in pts], marker="o", ms=3, label=tag)
            ax.set_xscale("log", base=2)
            ax.set_title(f"d = {d}, {'causal' if causal else 'non-causal'} (b4 h16 fp16)", fontsize=9)
            ax.set_xlabel("sequence length")
            ax.set_ylabel(ylabel, fontsize=8)
            ax.grid(alpha=0.3)
            ax.legend(fontsize=8)
        fig.tight_layout()
        fig.savefig(prefix + out, dpi=120)


if __name__ == "__main__":
    main()
