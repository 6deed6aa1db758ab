#!/usr/bin/env python3
"""|LSE(kernel) - LSE(fp32 math)| over EVERY row of a BASELINE-size problem, per kernel set, as a distribution (the forward's
16x16x32 fp16 kernel sums its softmax rows in the matrix pipe from the ROUNDED P values; this is the account of what that does to LSE).
The expectation is torch fp32: scores = (q k^T) / sqrt(d) per head in fp32, logsumexp in fp32 (its own error ~1e-6 |LSE|).
Usage: lse_error.py [--out gpurun_out/lse_error.json]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lse_error.json")); ap.add_argument("--only", default=""); a = ap.parse_args()
dev = torch.device("cuda:0")
CASES = {"c2 fwd b4 s4096 h32 d128 fp16": (4, 4096, 32, 128, False, 1.0), "c3 fwd b4 s16384 h32 d128 fp16 causal": (4, 16384, 32, 128, True, 1.0),
         "c5 shard b4 s16384 non-causal (heads 0..7 only checked)": (4, 16384, 32, 128, False, 1.0),
         "b2 s2048 h8 causal, inputs x6 (scores x36)": (2, 2048, 8, 128, True, 6.0),
         "b2 s777 h4 causal": (2, 777, 4, 128, True, 1.0),
         # round 5: bf16 (P rounds to 8 bits) - the BASELINE configs[3] forward, its causal twin, and the magnitude / short cases again
         "c4 fwd b4 s8192 h32 d128 bf16": (4, 8192, 32, 128, False, 1.0, "bf16"),
         "b4 s8192 h32 d128 bf16 causal": (4, 8192, 32, 128, True, 1.0, "bf16"),
         "b2 s16384 h8 d128 bf16 causal": (2, 16384, 8, 128, True, 1.0, "bf16"),
         "b2 s2048 h8 causal bf16, inputs x6 (scores x36)": (2, 2048, 8, 128, True, 6.0, "bf16"),
         "b2 s777 h4 causal bf16": (2, 777, 4, 128, True, 1.0, "bf16")}
if a.only:
    CASES = {k: v for k, v in CASES.items() if any(x in k for x in a.only.split(","))}
out = {"what": __doc__.split("\n")[0], "library": capi.lib().fa_build_info().decode(), "cases": {}}
for name, case in CASES.items():
    (b, s, h, d, causal, mag), dt = case[:6], (torch.bfloat16 if len(case) > 6 and case[6] == "bf16" else torch.float16)
    g = torch.Generator(device=dev).manual_seed(1234)
    q, k, v = ((torch.randn(b, s, h, d, device=dev, dtype=torch.float32, generator=g) * (mag if i < 2 else 1.0)).to(dt) for i in range(3))
    o = torch.empty_like(q); lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
    heads = range(h) if "heads 0..7" not in name else range(8)
    ref = torch.empty(b, len(heads), s, device=dev, dtype=torch.float32)
    mask = torch.ones(s, s, device=dev, dtype=torch.bool).tril_() if causal else None
    for bi in range(b):
        for j, hi in enumerate(heads):
            sc = (q[bi, :, hi].float() @ k[bi, :, hi].float().T) * (d ** -0.5)
            if causal:
                sc.masked_fill_(~mask, float("-inf"))
            ref[bi, j] = torch.logsumexp(sc, -1)
            del sc
    res = {}
    for pol, pname in ((capi.POLICY_MFMA32, "32x32x16 kernel (VALU row sums)"), (capi.POLICY_MFMA16, "16x16x32 kernel (fp16: MFMA row sums after the exact prefix)")):
        capi.set_kernel_policy(pol)
        capi.mha_fwd(q, k, v, o, lse, causal); torch.cuda.synchronize()
        err = (lse[:, list(heads)] - ref).abs()
        rel = err / ref.abs().clamp_min(1.0)
        e = err.flatten().double()
        qs = torch.quantile(e[torch.randperm(e.numel(), device=dev)[: min(e.numel(), 4_000_000)]], torch.tensor([0.5, 0.99, 0.9999], device=dev, dtype=torch.float64))
        rows = torch.arange(s, device=dev)
        buckets = {}
        for lo, hi_ in ((0, 64), (64, 256), (256, 1024), (1024, 4096), (4096, s)):
            if lo < s:
                sel = err[..., lo:min(hi_, s)]
                buckets[f"rows {lo}..{min(hi_, s) - 1}"] = {"max": sel.max().item(), "mean": sel.mean().item()}
        res[pname] = {"kernel": capi.kernel_name("fwd", b, s, s, h, d, causal, "bf16" if dt == torch.bfloat16 else "fp16"), "rows": int(e.numel()), "max_abs": e.max().item(), "mean_abs": e.mean().item(),
                      "median": qs[0].item(), "p99": qs[1].item(), "p99.99": qs[2].item(), "max_rel_to_max(|lse|,1)": rel.max().item(), "by_query_row": buckets}
    capi.set_kernel_policy(capi.POLICY_AUTO)
    out["cases"][name] = res
    print(name, json.dumps({k_: {"max_abs": v_["max_abs"], "p99.99": v_["p99.99"], "mean_abs": v_["mean_abs"]} for k_, v_ in res.items()}), flush=True)
    del q, k, v, o, lse, ref
    torch.cuda.empty_cache()
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
