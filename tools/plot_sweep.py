#!/usr/bin/env python3
"""Bar charts of the seqlen sweep in a bench.py JSON line, in the spirit of the reference README's figures (its utils/plot_kernels.py
draws speed-up over PyTorch and throughput per sequence length): speed-up of the forward over torch's fused SDPA on the same GPU
and forward TFLOP/s, causal and non-causal.  Usage: plot_sweep.py BENCH.json OUT_PREFIX  -> OUT_PREFIX_speedup.png, OUT_PREFIX_tflops.png"""
import json
import sys

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    with open(path) as f:
        line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    extra = json.loads(line)["extra"]
    sweeps = {"non-causal": extra.get("sweep_b4_h32_d128_fp16_noncausal", {}), "causal": extra.get("sweep_b4_h32_d128_fp16_causal", {})}
    seqs = sorted({int(s) for sw in sweeps.values() for s in sw})
    for key, ylabel, title, out in (("speedup_vs_torch_sdpa", "speed-up over torch SDPA (same MI355X)", "Forward, b4 h32 d128 fp16: speed-up over PyTorch-ROCm fused SDPA", "_speedup.png"),
                                    ("tflops", "TFLOP/s (4*b*h*sq*sk*d, causal x 1/2)", "Forward, b4 h32 d128 fp16: throughput", "_tflops.png")):
        fig, ax = plt.subplots(figsize=(8, 4.2))
        w = 0.38
        for i, (name, sw) in enumerate(sweeps.items()):
            ys = [sw.get(str(s), {}).get(key, 0.0) or 0.0 for s in seqs]
            xs = [j + (i - 0.5) * w for j in range(len(seqs))]
            bars = ax.bar(xs, ys, w, label=name)
            for x, y in zip(xs, ys):
                ax.text(x, y, f"{y:.2f}" if key.startswith("speedup") else f"{y:.0f}", ha="center", va="bottom", fontsize=7)
        ax.set_xticks(range(len(seqs)))
        ax.set_xticklabels([str(s) for s in seqs])
        ax.set_xlabel("sequence length")
        ax.set_ylabel(ylabel)
        ax.set_title(title, fontsize=10)
        if key.startswith("speedup"):
            ax.axhline(1.0, color="k", lw=0.6)
        ax.legend()
        fig.tight_layout()
        fig.savefig(prefix + out, dpi=130)
        print(prefix + out)


if __name__ == "__main__":
    main()
