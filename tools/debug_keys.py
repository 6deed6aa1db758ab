#!/usr/bin/env python3
"""Which keys does a forward build get wrong?  Q = 0 makes every visible key weigh 1 / n_visible; V = a tiled identity (key k -> column k % d) turns O into a
histogram of the keys each row saw.  Prints, for every row whose histogram is off, the key residues with a wrong weight (in units of one key).
Usage: debug_keys.py --lib tools/abl/libfa_x.so --sq 512 --sk 512 --causal 1 [--policy 1] [--dtype fp16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import torch  # noqa: E402
from flash_attn_turing import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--policy", type=int, default=1)
ap.add_argument("--sq", type=int, default=512)
ap.add_argument("--sk", type=int, default=512)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--max-rows", type=int, default=40)
a = ap.parse_args()
if a.lib:
    capi.LIBRARY_PATH = os.path.abspath(a.lib)
capi.set_kernel_policy(a.policy)
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
q = torch.zeros(1, a.sq, 1, a.d, device=dev, dtype=dt)
k = torch.randn(1, a.sk, 1, a.d, device=dev, dtype=dt)
v = torch.zeros(1, a.sk, 1, a.d, device=dev, dtype=dt)
v[0, torch.arange(a.sk), 0, torch.arange(a.sk) % a.d] = 1.0
o = torch.full_like(q, float("nan"))
lse = torch.full((1, 1, a.sq), float("nan"), device=dev, dtype=torch.float32)
capi.mha_fwd(q, k, v, o, lse, bool(a.causal))
torch.cuda.synchronize()
rows = torch.arange(a.sq, device=dev)
nvis = (rows + (a.sk - a.sq) + 1).clamp(0, a.sk) if a.causal else torch.full_like(rows, a.sk)
# expected histogram: keys 0 .. nvis-1 folded modulo d
full, rem = nvis // a.d, nvis % a.d
exp_cnt = full[:, None] + (torch.arange(a.d, device=dev)[None, :] < rem[:, None]).long()
got_cnt = o[0, :, 0, :].float() * nvis[:, None].float()
lse_exp = torch.log(nvis.float().clamp(min=1))
bad_rows = ((got_cnt - exp_cnt.float()).abs().amax(dim=1) > 0.25) | ((lse[0, 0] - lse_exp).abs() > 1e-2)
print(f"{int(bad_rows.sum())} of {a.sq} rows off")
shown = 0
for r in torch.nonzero(bad_rows).flatten().tolist():
    diff = got_cnt[r] - exp_cnt[r].float()
    cols = torch.nonzero(diff.abs() > 0.25).flatten().tolist()
    # implied row sum: l = exp(lse) in units of one key
    print(f"row {r:5d} visible {int(nvis[r]):5d}  exp(lse) {float(torch.exp(lse[0, 0, r])):9.2f}  residues off: " + " ".join(f"{c}:{float(diff[c]):+.1f}" for c in cols[:24])
          + (" ..." if len(cols) > 24 else ""))
    shown += 1
    if shown >= a.max_rows:
        break
