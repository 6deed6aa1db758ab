#!/usr/bin/env python3
"""Development aid: error structure inside one output tile for different V patterns."""
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
sq, sk, d = 256, 128, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, sq, 1, d, device=dev, dtype=torch.float16, generator=g)
k = torch.randn(1, sk, 1, d, device=dev, dtype=torch.float16, generator=g)
vr = torch.randn(1, sk, 1, d, device=dev, dtype=torch.float16, generator=g)
def run(v, tag):
    o = torch.empty_like(q); lse = torch.empty(1, 1, sq, device=dev, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, False)
    assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    o_r, _ = U.torch_attention_ref(q, k, v, None, False)
    e = (o.float() - o_r).abs()[0, :, 0]
    blk = e.view(8, 32, 4, 32).amax(dim=(1, 3))
    print(tag, "max err per (32-row block, d-block):", " | ".join(" ".join(f"{x:7.1e}" for x in r) for r in blk.tolist()))
    return e
run(vr[:, :, :, :1].expand(-1, -1, -1, d).contiguous(), "V random per key, same for all d:")
v2 = vr.clone(); v2[0, 64:] = 0
run(v2, "V random, only tile 0 keys nonzero:   ")
v3 = vr.clone(); v3[0, :64] = 0
run(v3, "V random, only tile 1 keys nonzero:   ")
e = run(vr, "V random:                             ")
t5 = e[32:64, 64:96]
print("tile (rows 32..63, cols 64..95): rows with error > 1e-2:", (t5.amax(1) > 1e-2).nonzero().flatten().tolist())
print("                                 cols with error > 1e-2:", (t5.amax(0) > 1e-2).nonzero().flatten().tolist())
for sl in range(8):
    v4 = torch.zeros_like(vr); v4[0, sl * 16:(sl + 1) * 16] = vr[0, sl * 16:(sl + 1) * 16]
    run(v4, f"V random, only 16-key slice {sl} nonzero: ")
