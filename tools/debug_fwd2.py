#!/usr/bin/env python3
"""Development aid: which 16-key slices reach which output column blocks?  V = indicator of one 16-key slice, so O[row, d] = softmax mass
of that slice for every d; columns of different 32-wide d-blocks must agree."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(sys.argv[1])
L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
dev = torch.device("cuda:0")
sq, sk, d = 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, sq, 1, d, device=dev, dtype=torch.float16, generator=g)
k = torch.randn(1, sk, 1, d, device=dev, dtype=torch.float16, generator=g)
print("rows = 16-key slice, cols = mass seen by d-block 0,1,2,3 for query row 40 (q-block 1 of wave 0) | same for row 8 (q-block 0)")
for sl in range(sk // 16):
    v = torch.zeros(1, sk, 1, d, device=dev, dtype=torch.float16)
    v[0, sl * 16:(sl + 1) * 16] = 1.0
    o = torch.empty_like(q); lse = torch.empty(1, 1, sq, device=dev, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, False)
    assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    r1 = [o[0, 40, 0, 32 * b + 5].item() for b in range(4)]
    r0 = [o[0, 8, 0, 32 * b + 5].item() for b in range(4)]
    print(f"slice {sl:2d} (tile {sl // 4}, ts {sl % 4}): " + " ".join(f"{x:7.4f}" for x in r1) + "   |   " + " ".join(f"{x:7.4f}" for x in r0))
