#!/usr/bin/env python3
"""Development aid: forward of a library build vs an fp32 torch reference, error broken down by 32-row block and 32-column block."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flash_attn_turing import capi
import _util as U
L = ctypes.CDLL(sys.argv[1])
L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
dev = torch.device("cuda:0")
CASES = [tuple(int(x) if i < 5 else (x == "1") for i, x in enumerate(c.split(","))) + (torch.float16,) for c in sys.argv[2:]]
for (b, sq, sk, h, d, causal, dt) in CASES or ((1, 64, 64, 1, 128, False, torch.float16), (1, 256, 64, 1, 128, False, torch.float16), (1, 256, 128, 1, 128, False, torch.float16),
                                  (1, 256, 256, 1, 128, False, torch.bfloat16), (1, 256, 256, 1, 128, True, torch.float16), (2, 512, 512, 2, 128, False, torch.float16)):
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(b, sq, h, d, device=dev, dtype=dt, generator=g)
    k = torch.randn(b, sk, h, d, device=dev, dtype=dt, generator=g)
    v = torch.randn(b, sk, h, d, device=dev, dtype=dt, generator=g)
    o = torch.full_like(q, float("nan")); lse = torch.full((b, h, sq), float("nan"), device=dev, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, causal)
    rc = L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o_r, lse_r = U.torch_attention_ref(q, k, v, None, causal)
    eo = (o.float() - o_r).abs()[0, :, 0]          # (sq, d) of batch 0 head 0
    el = (lse - lse_r).abs()[0, 0]
    print(f"rc={rc} b{b} sq{sq} sk{sk} causal={causal} {dt}: max|dO| {eo.max().item():.3e} (nan {torch.isnan(o).sum().item()}) max|dLSE| {el.max().item():.3e}")
    pad = (-sq) % 32
    eo = torch.nn.functional.pad(eo, (0, 0, 0, pad)); el = torch.nn.functional.pad(el, (0, pad))
    sq = sq + pad
    blk = eo.view(sq // 32, 32, 4, 32).amax(dim=(1, 3))
    print("   per 32-row block x 32-col block max err:\n   " + "\n   ".join(" ".join(f"{x:8.1e}" for x in r) for r in blk.tolist()))
    print("   LSE err per 32-row block: " + " ".join(f"{x:8.1e}" for x in el.view(sq // 32, 32).amax(1).tolist()))
